#!/usr/bin/env python3
"""Where the time inside a tail lock-step's kernels goes: the profiling build (make -C deep-neuroevolution_amd/csrc clock ->
libdne_hip_clock.so) leaves the 100 MHz wall clock at a few milestones of the LAST launch of k_env_render / k_conv12t / k_fc_tail /
k_tail_step (thread 0 of each of the first 128 workgroups).  Prints, per kernel, first start / last end relative to the lock-step's
first stamp and the mean time between milestones, in microseconds.
    DNE_LIB_PATH=.../libdne_hip_clock.so python tools/phase_clock.py 8"""
import ctypes as C, json, os, sys
import numpy as np
os.environ.setdefault("DNE_DEBUG_IMMORTAL", "1")
os.environ.setdefault("DNE_SPEC_MAX", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * pairs, ref_count=128)
noise = es.SharedNoiseTable(count=25_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
_, idx, seeds = es.generation_inputs(noise.noise.size, e.P, pairs, 3, 0, 1)
e.es_eval(idx, 0.02, 64, seeds)
e.es_eval(idx, 0.02, 64, seeds)
buf = np.zeros((6, 128, 8), np.int64)
fn = e.lib.dne_debug_phase_clock
fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
if fn(e.h, buf.ctypes.data_as(C.c_void_p)) != 0:
    raise SystemExit("this library has no phase clock: build it with make clock and set DNE_LIB_PATH")
names = {0: ("k_env_render", ["start", "tables+ram", "unique rows", "horizontal", "end"]),
         1: ("k_conv12t", ["start", "staged", "conv1", "conv2 weights", "end"]),
         2: ("k_fc_tail", ["start", "x staged", "streamed", "barrier", "end"]),
         3: ("k_tail_step", ["start", "operands", "out layer", "logits", "end"])}
ticks = 0.01   # 100 MHz -> microseconds
# the last lock-step ran conv12t -> fc_tail -> tail_step -> render; take the stamps of its kernels relative to conv12t's first start
t0 = min(int(buf[1, w, 0]) for w in range(128) if buf[1, w, 0] > 0)
out = {"pairs": pairs, "origin": "first k_conv12t workgroup start of the last lock-step"}
for k, (name, ms) in names.items():
    wgs = [w for w in range(128) if buf[k, w, 0] > 0 and buf[k, w, len(ms) - 1] >= buf[k, w, 0]]
    if not wgs:
        continue
    b = buf[k, wgs][:, :len(ms)].astype(np.float64)
    d = {"workgroups": len(wgs), "first_start_us": round((b[:, 0].min() - t0) * ticks, 2), "last_start_us": round((b[:, 0].max() - t0) * ticks, 2),
         "last_end_us": round((b[:, -1].max() - t0) * ticks, 2)}
    for i in range(1, len(ms)):
        seg = (b[:, i] - b[:, i - 1]) * ticks
        d["%s -> %s" % (ms[i - 1], ms[i])] = {"mean_us": round(float(seg.mean()), 2), "max_us": round(float(seg.max()), 2)}
    out[name] = d
print(json.dumps(out, indent=1))
