#!/usr/bin/env python3
"""Where a generation's lock-steps go: the episode-length distribution of one population share, the number of
lock-steps spent at each active-count level and the engine's stage timers."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=312)
ap.add_argument("--tslimit", type=int, default=5000)
ap.add_argument("--noise-count", type=int, default=250_000_000)
a = ap.parse_args()
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * a.pairs, ref_count=128, profile_events=False)
noise = es.SharedNoiseTable(count=a.noise_count); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
for rep in range(2):
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, a.pairs, rep, 0, 1)
    t = time.time(); ret, sg, ln = e.es_eval(idx, 0.02, a.tslimit, seeds); wall = time.time() - t
glen = ln.max(axis=1)                      # a pair stays in the active list until both members are done
steps = np.arange(glen.max())
active = (glen[None, :] > steps[:, None]).sum(1)
edges = [1, 4, 12, 24, 32, 48, 96, 200, 1000, 10 ** 9]
out = {"pairs": a.pairs, "wall_ms": wall * 1e3, "ref_ms": e.profile()["ref_ms"], "env_steps": int(ln.sum()),
       "max_len": int(glen.max()), "mean_len": float(ln.mean())}
lo = 0
for hi in edges:
    out["steps_active_%d_%d" % (lo + 1, hi)] = int(((active > lo) & (active <= hi)).sum())
    lo = hi
print(json.dumps(out))
