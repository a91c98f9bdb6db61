#!/usr/bin/env python3
"""Where a k_conv12 workgroup's time goes at full width (profiling build: make -C deep-neuroevolution_amd/csrc clock): thread 0 of workgroups
1024 .. 1151 of the LAST launch stamps the 100 MHz wall clock at: start, image + conv1 weights staged (barrier), conv1 done, conv2 weights
formed (barrier), conv2's MFMAs done, end.  Per setting (knobs as tools/ab_inproc.py; "DNE_NSUB=1" = one window: the kernel alone).
    DNE_LIB_PATH=.../libdne_hip_clock.so python tools/conv12_phase_clock.py "DNE_NSUB=1" "X=0" """
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

noise = es.SharedNoiseTable()
ref = None
names = ["start", "staged", "conv1", "conv2 weights", "conv2", "end"]
for st in sys.argv[1:] or ["DNE_NSUB=1", "X=0"]:
    env = dict(kv.split("=") for kv in st.split())
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = _lib.Engine(_lib.KIND_ES, 18, max_members=5000, ref_count=128)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    noise.attach(e)
    e.set_theta(policies.xavier_flat(18, 0))
    if ref is None:
        henv = policies.HipAtariEnv(e, seed=0)
        ref = np.rint(np.stack(es.get_ref_batch(henv, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
    e.set_ref_batch(ref)
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, 2500, 1, 0, 1)
    e.es_eval(idx, 0.02, 12, seeds)     # nobody dies in 12 lock-steps: every launch is full width
    buf = np.zeros((6, 128, 8), np.int64)
    fn = e.lib.dne_debug_phase_clock
    fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
    if fn(e.h, buf.ctypes.data_as(C.c_void_p)) != 0:
        raise SystemExit("this library has no phase clock: build it with make clock and set DNE_LIB_PATH")
    b = buf[4][:, :6].astype(np.float64)
    ok = (b[:, 0] > 0) & (np.diff(b, axis=1) >= 0).all(axis=1)
    b = b[ok]
    out = {"setting": st, "workgroups": int(ok.sum()), "workgroup_us_mean": round(float((b[:, 5] - b[:, 0]).mean()) * 0.01, 2)}
    for i in range(1, 6):
        seg = (b[:, i] - b[:, i - 1]) * 0.01
        out["%s -> %s" % (names[i - 1], names[i])] = {"mean_us": round(float(seg.mean()), 2), "p90_us": round(float(np.percentile(seg, 90)), 2)}
    print(json.dumps(out))
    e.close()
