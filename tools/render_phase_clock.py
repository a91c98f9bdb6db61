#!/usr/bin/env python3
"""Where a k_env_render workgroup's time goes at full width (one workgroup per member, DNE_RENDER_THREADS threads): the profiling build's
milestones (thread 0 of the first 128 workgroups of the LAST launch) -- start, tables + RAM in LDS, unique rows painted, horizontal pass, end.
    DNE_LIB_PATH=.../libdne_hip_clock.so python tools/render_phase_clock.py [pairs] [knob=value ...]"""
import ctypes as C, json, os, sys
import numpy as np
os.environ.setdefault("DNE_DEBUG_IMMORTAL", "1")
os.environ.setdefault("DNE_NSUB", "1")
for kv in sys.argv[2:]:
    k, v = kv.split("="); os.environ[k] = v
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * pairs, ref_count=128)
noise = es.SharedNoiseTable(count=25_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
_, idx, seeds = es.generation_inputs(noise.noise.size, e.P, pairs, 3, 0, 1)
e.es_eval(idx, 0.02, 12, seeds)
buf = np.zeros((6, 128, 8), np.int64)
fn = e.lib.dne_debug_phase_clock
fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
if fn(e.h, buf.ctypes.data_as(C.c_void_p)) != 0:
    raise SystemExit("this library has no phase clock: build it with make clock and set DNE_LIB_PATH")
ms = ["start", "tables+ram", "unique rows", "horizontal", "end"]
b = buf[0]
ok = [w for w in range(128) if b[w, 0] > 0 and b[w, 4] >= b[w, 0]]
d = np.diff(b[ok][:, :5], axis=1) * 0.01
fine = {"keys claimed": round(float(((b[ok][:, 5] - b[ok][:, 1]) * 0.01).mean()), 2), "rows described": round(float(((b[ok][:, 6] - b[ok][:, 5]) * 0.01).mean()), 2),
        "rows painted": round(float(((b[ok][:, 2] - b[ok][:, 6]) * 0.01).mean()), 2)} if (b[ok][:, 5] > 0).all() and (b[ok][:, 6] > 0).all() else None
print(json.dumps({"pairs": pairs, "unique_rows_split_us": fine, "knobs": sys.argv[2:], "workgroups": len(ok), "phase_us_mean": {ms[i + 1]: round(float(d[:, i].mean()), 2) for i in range(4)},
                  "phase_us_p90": {ms[i + 1]: round(float(np.percentile(d[:, i], 90)), 2) for i in range(4)},
                  "total_us_mean": round(float(d.sum(1).mean()), 2), "span_first_start_to_last_end_us": round(float((b[ok][:, 4].max() - b[ok][:, 0].min()) * 0.01), 1)}))
