#!/usr/bin/env python3
"""Deep GA at FIXED width: `--members` children of 20 parents (chains of two seeds), every member kept alive (DNE_DEBUG_IMMORTAL) for
`--tslimit` lock-steps -- the GA counterpart of tools/kbench.py for kernel traces and counter passes."""
import argparse, json, os, sys, time
import numpy as np
os.environ.setdefault("DNE_DEBUG_IMMORTAL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es

ap = argparse.ArgumentParser()
ap.add_argument("--members", type=int, default=1000)
ap.add_argument("--tslimit", type=int, default=8)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
noise = es.SharedNoiseTable()
e = _lib.Engine(_lib.KIND_GA, 18, max_members=a.members)
noise.attach(e)
rs = np.random.RandomState(0)
parents = [[int(noise.sample_index(rs, e.P))] for _ in range(20)]
kids = [parents[rs.randint(20)] + [int(noise.sample_index(rs, e.P))] for _ in range(a.members)]
seeds = rs.randint(0, 2 ** 32, size=a.members, dtype=np.uint64).astype(np.uint32)
for rep in range(a.reps):
    t = time.time(); ret, _, ln = e.ga_eval(kids, 0.005, a.tslimit, seeds); wall = time.time() - t
    print(json.dumps({"rep": rep, "members": a.members, "wall_ms": 1e3 * wall, "ms_per_lock_step": 1e3 * wall / a.tslimit, "env_steps": int(ln.sum())}))
