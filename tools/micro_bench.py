#!/usr/bin/env python3
"""The env-free kernel micro-benchmarks of SURVEY.md section 8(d) with its inputs (wall clock around the C-ABI calls,
device buffers only -- copy_out=False -- unless the call returns a small result):
aggregate N in {128, 2500}; materialise; step forward B in {256, 5000}; reference pass; GA rebuild chains of 1 / 10 / 100 /
259 seeds; novelty against archives of 3 / 32 / 100 trajectories; centered ranks and GA selection on tie-heavy returns."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

out = {}
noise = es.SharedNoiseTable(count=250_000_000)
e = _lib.Engine(_lib.KIND_ES, 18, max_members=5000, ref_count=128)
noise.attach(e)
P = e.P
e.set_theta(policies.xavier_flat(18, 0))


def timed(f, reps=5):
    f()
    t = time.time()
    for _ in range(reps):
        f()
    return (time.time() - t) / reps * 1e3


for N in (128, 2500):
    rs = np.random.RandomState(0)
    idx = np.array([rs.randint(0, 250_000_000 - P + 1) for _ in range(N)], np.int64)
    ret = (10 * np.random.RandomState(1).poisson(20, (N, 2))).astype(np.float32)
    w = e.centered_ranks(ret).reshape(N, 2)
    ms = timed(lambda: e.weighted_sum(idx, w[:, 0] - w[:, 1], 2 * N, copy_out=False))
    out["aggregate_N%d" % N] = {"ms": ms, "GB_per_s_algorithmic": N * 4 * P / ms / 1e6}
    out["centered_ranks_2N%d" % (2 * N)] = {"ms": timed(lambda: e.centered_ranks(ret))}
    ms = timed(lambda: e.materialize(idx[:min(N, 512)], 0.02, copy_out=False))
    out["materialise_%d_pairs" % min(N, 512)] = {"ms": ms, "GB_per_s": min(N, 512) * 12 * P / ms / 1e6}
out["ga_select_1020_top20"] = {"ms": timed(lambda: e.ga_select((10 * np.random.RandomState(1).poisson(20, 1020)).astype(np.float32), 20))}

env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
for B in (256, 5000):
    obs = np.random.RandomState(2).randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
    rs = np.random.RandomState(0)
    idx = np.array([rs.randint(0, 250_000_000 - P + 1) for _ in range(B // 2)], np.int64)
    e.set_members(np.zeros(B, np.int32), np.repeat(idx, 2), np.tile(np.array([0.02, -0.02], np.float32), B // 2))
    e.env_reset(np.arange(B, dtype=np.uint32))
    ms_ref = timed(lambda: e.ref_pass(B), reps=2)
    e.env_set_observation(obs)
    ms = timed(lambda: e.act(B), reps=5)
    out["step_forward_B%d" % B] = {"ms": ms, "note": "dne_act on explicit members (one window, no pair sharing, copies actions + logits out)",
                                   "GB_per_s_algorithmic": B * (4 * P + 28224) / ms / 1e6}
    out["ref_pass_B%d" % B] = {"ms": ms_ref, "TFLOP_per_s": B * 0.971e9 / ms_ref / 1e9}
e.close()

g = _lib.Engine(_lib.KIND_GA, 18, max_members=8)
noise.attach(g)
rs = np.random.RandomState(3)
for L in (1, 10, 100, 259):
    seeds = rs.randint(0, 250_000_000 - g.P + 1, L).astype(np.int64)
    ms = timed(lambda: g.ga_rebuild(0, seeds, 0.002, copy_out=False), reps=3)
    out["ga_rebuild_chain_%d" % L] = {"ms": ms, "GB_per_s": (1 + L) * 4 * g.P / ms / 1e6}
for A in (3, 32, 100):
    rs = np.random.RandomState(4)
    arch = [rs.randint(0, 256, (int(rs.randint(200, 5001)), 128)).astype(np.uint8) for _ in range(A)]
    bc = rs.randint(0, 256, (int(rs.randint(200, 5001)), 128)).astype(np.uint8)
    ms = timed(lambda: g.novelty(arch, bc, 10), reps=3)
    out["novelty_archive_%d" % A] = {"ms": ms, "archive_MB": sum(a.nbytes for a in arch) / 1e6, "note": "archive uploaded per call"}
g.close()
print(json.dumps(out, indent=1))
