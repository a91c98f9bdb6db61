import sys, os, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies
e = _lib.Engine(_lib.KIND_ES, 18, max_members=4, ref_count=128)
noise = es.SharedNoiseTable(count=30_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
_, idx, seeds = es.generation_inputs(noise.noise.size, e.P, 2, 0, 0, 1)
e.es_eval(idx, 0.02, 50, seeds)
ts = (C.c_longlong * 32)()
e.lib.dne_debug_ts(ts)
t = np.array(ts[:16], dtype=np.int64)
names = {0: "start", 1: "out loads+tables", 2: "out chain", 3: "argmax+logic", 8: "observe entry"}
for k in range(1, 16):
    if t[k] and t[k - 1]: print(k, names.get(k, ""), (t[k] - t[k - 1]) * 10, "ns")
print("total", (t[15] - t[0]) * 10, "ns")
t2 = np.array(ts[:32], dtype=np.int64)
print("nu", t2[20], "pixstate", (t2[4] - t2[10]) * 10, "loop(wave0)", (t2[5] - t2[4]) * 10, "wait", (t2[11] - t2[5]) * 10)
