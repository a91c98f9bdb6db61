#!/usr/bin/env python3
"""Config 4 on one GPU: the NS-ES meta-population loop at pop 5000 (tools/workloads.py:nses) -- rollouts record their
RAM trajectories in HBM, dne_novelty_batch scores all 5000 against the device-resident archive, blend + update, next parent."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as W
from dne_hip import es

ap = argparse.ArgumentParser()
ap.add_argument("--archive", type=int, default=32); ap.add_argument("--pop", type=int, default=5000); ap.add_argument("--iterations", type=int, default=2)
a = ap.parse_args()
r = W.nses(es.SharedNoiseTable(), iterations=a.iterations, pop=a.pop, archive_extra=max(a.archive - 3, 0))
r.pop("_cpu_inputs", None)   # numpy arrays for bench.py's cpu_baseline leg, not part of the report
for it in r.pop("iterations"):
    print(json.dumps(it))
print(json.dumps(r))
