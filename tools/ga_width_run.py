#!/usr/bin/env python3
"""One Deep-GA evaluation at FIXED width (every member kept alive: DNE_DEBUG_IMMORTAL) for T lock-steps -- the thing to put under
rocprofv3 --kernel-trace --stats when a width class of tools/ga_lockstep_profile.py needs explaining.
    DNE_DEBUG_IMMORTAL=1 python tools/ga_width_run.py 250 64 [--large]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
os.environ.setdefault("DNE_DEBUG_IMMORTAL", "1")
from dne_hip import _lib, es, ga_gpu
W, T = int(sys.argv[1]), int(sys.argv[2])
LARGE = "--large" in sys.argv
noise = es.SharedNoiseTable(count=int(os.environ.get("NOISE_COUNT", "60000000")))
e = _lib.Engine(_lib.KIND_GA_LARGE if LARGE else _lib.KIND_GA, 18, max_members=W)
noise.attach(e)
rs = np.random.RandomState(0)
parents = [[int(noise.sample_index(rs, e.P))] for _ in range(20)]
kids = [parents[rs.randint(20)] + [int(noise.sample_index(rs, e.P))] for _ in range(W)]
seeds = rs.randint(0, 2 ** 32, size=W, dtype=np.uint64).astype(np.uint32)
if LARGE:
    e.ga_set_init_scale(ga_gpu.model_scale_by(18, _lib.KIND_GA_LARGE))
    run = lambda t: e.ga_eval_powers([(c[0],) + tuple((s, 0.002) for s in c[1:]) for c in kids], t, seeds)
else:
    run = lambda t: e.ga_eval(kids, 0.005, t, seeds)
run(8)
t0 = time.time(); run(T); wall = time.time() - t0
print(json.dumps({"width": W, "steps": T, "wall_ms": 1e3 * wall, "us_per_lock_step_incl_setup": 1e6 * wall / T}))
