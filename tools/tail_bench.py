#!/usr/bin/env python3
"""Latency of one lock-step at small active counts: `pairs` antithetic pairs that all live exactly `--steps` steps
(DNE_DEBUG_IMMORTAL=1: game over does not end an episode -- timing only), wall time / steps.  The list of pair counts spans
the speculative tail (<= 4 pairs), the quad fc (<= 24) and the column-split fc (<= 96).
    DNE_DEBUG_IMMORTAL=1 python tools/tail_bench.py [1,2,4,8,16,24,48,96] [--steps 200]"""
import json, os, sys, time
import numpy as np
os.environ.setdefault("DNE_DEBUG_IMMORTAL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 208      # a multiple of the 16-step burst
counts = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else [1, 2, 4, 8, 16, 24, 48, 96]
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * max(counts), ref_count=128)
noise = es.SharedNoiseTable(count=25_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
out = {"steps": steps, "immortal": os.environ.get("DNE_DEBUG_IMMORTAL")}
for pairs in counts:
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, pairs, 3, 0, 1)
    e.es_eval(idx, 0.02, 32, seeds)
    walls = []
    for rep in range(3):
        t = time.time(); ret, sg, ln = e.es_eval(idx, 0.02, steps, seeds); walls.append(time.time() - t)
    # the reference pass and the reset are part of an evaluation: subtract a zero-step baseline measured the same way
    t = time.time(); e.es_eval(idx, 0.02, 16, seeds); w16 = time.time() - t
    t = time.time(); e.es_eval(idx, 0.02, 16, seeds); w16 = min(w16, time.time() - t)
    us = 1e6 * (min(walls) - w16) / (steps - 16)
    out["pairs_%d" % pairs] = {"us_per_lock_step": round(us, 1), "all_alive": bool(ln.min() == steps), "eval_ms": round(1e3 * min(walls), 2)}
print(json.dumps(out))
