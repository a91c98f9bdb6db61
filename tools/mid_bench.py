import json, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies
pairs = int(sys.argv[1]); T = int(sys.argv[2])
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * pairs, ref_count=128)
noise = es.SharedNoiseTable(count=250_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
_, idx, seeds = es.generation_inputs(noise.noise.size, e.P, pairs, 0, 0, 1)
e.es_eval(idx, 0.02, T, seeds)
e.ref_pass(2 * pairs)
t = time.time(); e.ref_pass(2 * pairs); tr = time.time() - t
t = time.time(); ret, sg, ln = e.es_eval(idx, 0.02, T, seeds); wall = time.time() - t
print(json.dumps({"pairs": pairs, "T": T, "wall_ms": wall * 1e3, "ref_ms": tr * 1e3, "ms_per_step": (wall - tr) * 1e3 / T}))
