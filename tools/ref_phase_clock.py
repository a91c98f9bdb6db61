#!/usr/bin/env python3
"""Where the time inside the reference pass's k_conv2_ref / k_fc_ref goes: the profiling build (make -C deep-neuroevolution_amd/csrc
clock -> libdne_hip_clock.so) sums, on thread 0 of 128 workgroups spread evenly over the last launch, the 100 MHz wall clock between the
kernels' phase marks (forward.h: DNE_ACC).  Prints the mean per workgroup in microseconds and each phase's share.
    DNE_LIB_PATH=.../libdne_hip_clock.so REF_CHUNK=5000 python tools/ref_phase_clock.py"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
e.ref_pass(5000); e.ref_pass(5000)
buf = np.zeros((6, 128, 8), np.int64)
fn = e.lib.dne_debug_phase_clock
fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
if fn(e.h, buf.ctypes.data_as(C.c_void_p)) != 0:
    raise SystemExit("this library has no phase clock: build it with make clock and set DNE_LIB_PATH")
names = {4: ("k_conv2_ref<8> (8 frames per workgroup)", ["staging", "barrier", "MFMAs + previous frame's epilogue", "barrier + frame moments"]),
         5: ("k_fc_ref<4> (one quarter, 64 frames)", ["first stage: MFMAs + load issue + store of the next unit", "barrier", "next unit's operands requested", "second stage: MFMAs", "fold"])}
out = {}
for k, (name, ph) in names.items():
    m = buf[k, :, :len(ph)].astype(np.float64).mean(axis=0) * 0.01
    t0 = buf[k, :, 7].min()
    out[name] = {"us_per_workgroup": round(float(m.sum()), 2), "phases_us": {p: round(float(v), 2) for p, v in zip(ph, m)},
                 "share": {p: round(float(v / m.sum()), 3) for p, v in zip(ph, m)},
                 "samples_over_the_launch": {"start_ms": [round(float(buf[k, w, 7] - t0) * 1e-5, 2) for w in range(0, 128, 8)],
                                             "duration_us": [round(float(buf[k, w, 6] - buf[k, w, 7]) * 0.01, 1) for w in range(0, 128, 8)]}}
print(json.dumps(out, indent=1))
