#!/usr/bin/env python3
"""The reference pass of 5000 members (virtual batch norm: 128 reference frames through every member's perturbed network)
alone: ms per pass, mean of three.  REF_CHUNK=5000 runs it as ONE chunk (the kernels then do not overlap: under rocprofv3 their
durations are their own); default = the product path (chunks of 512 members alternating between two streams).
    python tools/ref_bench.py;  REF_CHUNK=5000 rocprofv3 --kernel-trace --stats -- python tools/ref_bench.py"""
import json, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies
e = _lib.Engine(_lib.KIND_ES, 18, max_members=5000, ref_count=128, ref_chunk=int(os.environ.get("REF_CHUNK", "0")))
noise = es.SharedNoiseTable(count=250_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
rs = np.random.RandomState(0)
idx = np.array([rs.randint(0, 250_000_000 - e.P + 1) for _ in range(2500)], np.int64)
e.set_members(np.zeros(5000, np.int32), np.repeat(idx, 2), np.tile(np.array([0.02, -0.02], np.float32), 2500))
e.ref_pass(5000)
t = time.time()
N = int(os.environ.get("REF_REPS", "3"))
for _ in range(N): e.ref_pass(5000)
print(json.dumps({"ref_pass_ms": (time.time() - t) / N * 1e3, "knobs": {k: v for k, v in os.environ.items() if k.startswith("DNE_")}}))
