#!/usr/bin/env python3
"""Config 3: Deep GA, 1000 children per generation, top-20 parents -- tools/workloads.py:ga_small / ga_large (--large:
the GPU tree's protocol on its LargeModel) on one GPU, one JSON line per generation + the summary bench.py puts under extra."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as W
from dne_hip import es

noise = es.SharedNoiseTable()
r = W.ga_large(noise) if "--large" in sys.argv else W.ga_small(noise)
for g in r.pop("generations"):
    print(json.dumps(g))
print(json.dumps(r))
