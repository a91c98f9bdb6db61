#!/usr/bin/env python3
"""Where a Deep-GA generation's time goes (BASELINE config 3: 1000 children, top-20 parents): (1) the cost of one lock-step at
fixed width -- every member kept alive (DNE_DEBUG_IMMORTAL) -- for 1000, 500, 250, 100, 48, 24, 8, 2 members; (2) how many
lock-steps of a real generation run at which width (the episode-length distribution of generation 1); (3) their product: the
generation's time by width class, next to the measured wall time.  --large: the GPU tree's LargeModel."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
LARGE = "--large" in sys.argv
WIDTHS = (1000, 500, 250, 100, 48, 24, 8, 2)
from dne_hip import _lib, es, ga, ga_gpu


def make(immortal):
    if immortal:
        os.environ["DNE_DEBUG_IMMORTAL"] = "1"
    else:
        os.environ.pop("DNE_DEBUG_IMMORTAL", None)
    e = _lib.Engine(_lib.KIND_GA_LARGE if LARGE else _lib.KIND_GA, 18, max_members=1000)
    noise.attach(e)
    if LARGE:
        e.ga_set_init_scale(ga_gpu.model_scale_by(18, _lib.KIND_GA_LARGE))
    return e


def evaluate(e, chains, tslimit, seeds):
    if LARGE:
        return e.ga_eval_powers([(c[0],) + tuple((s, 0.002) for s in c[1:]) for c in chains], tslimit, seeds)
    return e.ga_eval(chains, 0.005, tslimit, seeds)


noise = es.SharedNoiseTable()
rs = np.random.RandomState(0)
out = {"model": "LargeModel" if LARGE else "GAAtariPolicy network", "bytes_per_member_step": (4 * 4052658 if LARGE else 4 * 1008450) + 28224}
# ---- a real generation 0 + 1: parents, lengths, wall
e = make(False)
roots = [[int(noise.sample_index(rs, e.P))] for _ in range(1000)]
seeds = rs.randint(0, 2 ** 32, size=1000, dtype=np.uint64).astype(np.uint32)
ret, _, ln0 = evaluate(e, roots, 5000, seeds)
parents = [roots[i] for i in e.ga_select(ret, 20)]
kids = [parents[rs.randint(20)] + [int(noise.sample_index(rs, e.P))] for _ in range(1000)]
seeds = rs.randint(0, 2 ** 32, size=1000, dtype=np.uint64).astype(np.uint32)
evaluate(e, kids, 5000, seeds)
t0 = time.time(); ret, _, ln = evaluate(e, kids, 5000, seeds); wall = time.time() - t0
out["generation_1"] = {"wall_ms": 1e3 * wall, "env_steps": int(ln.sum()), "steps_per_s": float(ln.sum() / wall), "mean_len": float(ln.mean()),
                       "max_len": int(ln.max())}
steps = np.arange(ln.max())
active = (ln[None, :] > steps[:, None]).sum(1)            # members alive at each lock-step
e.close(); noise._engines.clear()
# ---- cost of a lock-step at fixed width
e = make(True)
cost = {}
for w in WIDTHS:
    sub, sd = kids[:w], seeds[:w]
    evaluate(e, sub, 32, sd)
    t = []
    for T in (48, 16):
        a = time.time(); evaluate(e, sub, T, sd); b = time.time(); evaluate(e, sub, T, sd); t.append(min(b - a, time.time() - b))
    cost[w] = 1e6 * (t[0] - t[1]) / 32
out["lock_step_us_at_width"] = {str(w): round(cost[w], 1) for w in WIDTHS}
out["lock_step_GBps_algorithmic_at_width"] = {str(w): round(w * out["bytes_per_member_step"] / cost[w] / 1e3, 1) for w in WIDTHS}
# ---- the generation by width class (cost interpolated linearly in the width between the measured points)
ws = np.array(sorted(WIDTHS), float); cs = np.array([cost[int(w)] for w in ws])
est = np.interp(active, ws, cs)
edges = [0, 4, 24, 96, 250, 500, 1000]
rows = []
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (active > lo) & (active <= hi)
    rows.append({"members_alive": "%d-%d" % (lo + 1, hi), "lock_steps": int(m.sum()), "env_steps": int(active[m].sum()), "estimated_ms": round(float(est[m].sum()) / 1e3, 1)})
out["generation_1_by_width"] = rows
out["estimated_total_ms"] = round(float(est.sum()) / 1e3, 1)
print(json.dumps(out, indent=1))
