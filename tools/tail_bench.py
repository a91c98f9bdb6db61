#!/usr/bin/env python3
"""Latency of one lock-step in the tail of a generation: a handful of pairs, wall time / number of lock-steps
(the longest episode).  The knobs DNE_FLOW_MAX (0 = five launches per lock-step, default = one k_step_flow launch) and
DNE_RENDER_BANDS select the variants."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

e = _lib.Engine(_lib.KIND_ES, 18, max_members=64, ref_count=128)
noise = es.SharedNoiseTable(count=25_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
out = {}
for pairs in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 2, 4, 8, 16, 24)):
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, pairs, 3, 0, 1)
    e.es_eval(idx, 0.02, 50, seeds)
    t = time.time(); ret, sg, ln = e.es_eval(idx, 0.02, 400, seeds); wall = time.time() - t
    out["pairs_%d" % pairs] = {"us_per_lock_step": round(1e6 * wall / ln.max(), 1), "lock_steps": int(ln.max()), "env_steps": int(ln.sum())}
print(json.dumps(out))
