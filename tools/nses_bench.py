#!/usr/bin/env python3
"""Config 4 on one GPU: NS-ES evaluation at pop 5000 -- rollouts record their RAM trajectories in HBM,
dne_novelty_batch scores all 5000 of them against an archive of `--archive` behaviour characterisations."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, nses, policies

ap = argparse.ArgumentParser(); ap.add_argument("--archive", type=int, default=32); ap.add_argument("--pairs", type=int, default=2500)
a = ap.parse_args()
tsl = 5000
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * a.pairs, ref_count=128, record_bc=True, bc_max_steps=tsl)
noise = es.SharedNoiseTable(); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
archive = []
for p in range(a.archive):     # archive = mean BCs of differently initialised parents (nses.py:95-117)
    e.set_theta(policies.xavier_flat(18, 100 + p))
    archive.append(nses.get_mean_bc(e, tsl, 1000 + p))
e.set_theta(policies.xavier_flat(18, 0))
print("archive trajectory lengths:", [len(x) for x in archive][:12], "...")
for gen in range(2):
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, a.pairs, gen, 0, 1)
    t0 = time.time(); ret, sg, ln = e.es_eval(idx, 0.02, tsl, seeds); t1 = time.time()
    nov = e.novelty_batch(archive, ln, 10); t2 = time.time()
    print(json.dumps({"gen": gen, "eval_s": round(t1 - t0, 3), "novelty_s": round(t2 - t1, 3), "env_steps": int(ln.sum()),
                      "steps_per_s_incl_novelty": round(ln.sum() / (t2 - t0)), "novelty_mean": float(nov.mean()),
                      "archive": a.archive}))
