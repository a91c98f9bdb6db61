#!/usr/bin/env python3
"""Per-stage cost of one lock-step at full width: every member stays alive for `tslimit` steps."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=2500)
ap.add_argument("--tslimit", type=int, default=24)
ap.add_argument("--noise-count", type=int, default=250_000_000)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--ref-chunk", type=int, default=0)
ap.add_argument("--sort-idx", action="store_true", help="members in ascending noise-index order (Infinity Cache locality experiment)")
ap.add_argument("--idx-range", type=int, default=0, help="draw the noise indices from [0, N) only: the whole working set cache-resident")
a = ap.parse_args()
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * a.pairs, ref_count=128, profile_events=True, ref_chunk=a.ref_chunk)
noise = es.SharedNoiseTable(count=a.noise_count); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
for rep in range(a.reps):
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, a.pairs, rep, 0, 1)
    if a.idx_range:
        idx = np.random.RandomState(rep).randint(0, a.idx_range, size=len(idx)).astype(idx.dtype)
    if a.sort_idx:
        idx = np.sort(idx)
    t = time.time(); ret, sg, ln = e.es_eval(idx, 0.02, a.tslimit, seeds); wall = time.time() - t
    p = e.profile()
    n = p["fc_launches"]
    print(json.dumps({"rep": rep, "wall_s": wall, "steps": int(ln.sum()), "launches": n,
                      "per_step_ms": {k: p[k] / n for k in ("conv_ms", "fc_ms", "env_ms")}, "ref_ms": p["ref_ms"],
                      "fc_GBps_alg": p["fc_full_units"] * 4064456 / (max(p["fc_full_ms"], 1e-9) * 1e-3) / 1e9,
                      "step_wall_ms": (wall * 1e3 - p["ref_ms"]) / a.tslimit}))   # every member lives through tslimit here
t = time.time(); g = e.weighted_sum(idx, np.random.RandomState(0).randn(len(idx)).astype(np.float32), 2 * len(idx), copy_out=False)
print("weighted_sum ms", e.profile()["reduce_ms"], "GB/s", len(idx) * e.P * 4 / e.profile()["reduce_ms"] / 1e6)
e.materialize(idx[:256], 0.02, copy_out=False)
print("materialize(256 pairs) ms", e.profile()["materialize_ms"], "GB/s", 256 * e.P * 12 / e.profile()["materialize_ms"] / 1e6)
