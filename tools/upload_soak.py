"""Diagnosis of the round-1 bench fault: repeat the one large pageable host -> device transfer of a run (the 1 GB noise
table) many times, with the runtime's own path for large pageable buffers (DNE_STAGED_COPY=0) or the engine's pinned
staging, under torch's bundled HIP runtime (--torch-first) or /opt/rocm's.  Each upload is verified at three places."""
import argparse
import os
import sys
import time

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--torch-first", action="store_true")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--count", type=int, default=250_000_000)
args = ap.parse_args()
if args.torch_first:
    import torch
    torch.cuda.set_device(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deep-neuroevolution_amd"))
from dne_hip import _lib
e = _lib.Engine(_lib.KIND_ES, 18, max_members=5000, ref_count=128)
t0 = time.time()
for it in range(args.iters):
    a = np.empty(args.count, np.float32)              # a fresh mapping every time, first touched just before the copy
    chunk = 1 << 24
    for i in range(0, args.count, chunk):
        a[i:i + chunk] = np.float32(it + 1) + np.arange(min(chunk, args.count - i), dtype=np.float32) * np.float32(1e-6)
    e.noise_upload(a)
    for pos in (0, args.count // 2 + 12345, args.count - 1000):
        assert np.array_equal(e.noise_get(pos, 1000), a[pos:pos + 1000]), (it, pos)
    del a
    sys.stderr.write("upload %d ok (%.1f s)\n" % (it, time.time() - t0)); sys.stderr.flush()
e.close()
print("upload_soak: %d uploads ok, torch_first=%s staged=%s" % (args.iters, args.torch_first, os.environ.get("DNE_STAGED_COPY", "1")))
