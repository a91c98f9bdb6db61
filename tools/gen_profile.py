#!/usr/bin/env python3
"""Where a generation's milliseconds go on a TRAINED theta (the bench's timed generations are 5..24, not 0): the headline ES runs `--gens`
generations, then generation `--gens` is evaluated as one rank of `--world` (its round-robin shard, es.generation_inputs) three ways --
the whole evaluation, the evaluation cut at the lock-step where at most 96 pairs are left (the tail kernels' range), and cut where fewer
than `mid_lo` pairs are left -- so the bulk, the mid range and the tail are MEASURED (wall-clock differences of the same evaluation), plus the
lock-step counts per active-width bucket.  One JSON line per world size.
    python tools/gen_profile.py --gens 15 --worlds 1,2,4,8"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

ap = argparse.ArgumentParser()
ap.add_argument("--gens", type=int, default=15)
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--pairs", type=int, default=2500)
ap.add_argument("--tslimit", type=int, default=5000)
ap.add_argument("--noise-count", type=int, default=250_000_000)
a = ap.parse_args()
CFG = es.Config(l2coeff=0.005, noise_stdev=0.02, episodes_per_batch=2 * a.pairs, timesteps_per_batch=10000, calc_obstat_prob=0.0,
                eval_prob=0.0, snapshot_freq=0, return_proc_mode="centered_rank", episode_cutoff_mode=5000)
OPT = {"type": "adam", "args": {"stepsize": 0.01}}
noise = es.SharedNoiseTable(count=a.noise_count)
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * a.pairs, ref_count=128, profile_events=False)
noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref); e.optimizer_reset()
for g in range(a.gens):
    es.es_generation(e, noise.noise.size, CFG, a.pairs, g, a.tslimit, OPT)
theta = e.get_theta()


def timed(eng, idx, seeds, T, reps=3):
    best = 1e9
    for _ in range(reps):
        t = time.time(); r = eng.es_eval(idx, 0.02, T, seeds); best = min(best, time.time() - t)
    return best * 1e3, r


for w in [int(x) for x in a.worlds.split(",")]:
    mine, idx, seeds = es.generation_inputs(noise.noise.size, e.P, a.pairs, a.gens, 0, w)
    eng = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * len(mine), ref_count=128, profile_events=False)
    noise.attach(eng); eng.set_theta(theta); eng.set_ref_batch(ref)
    full_ms, (ret, sg, ln) = timed(eng, idx, seeds, a.tslimit)
    ref_ms = eng.profile()["ref_ms"]
    glen = ln.max(axis=1)
    steps = np.arange(glen.max())
    active = (glen[None, :] > steps[:, None]).sum(1)
    out = {"world": w, "pairs": len(mine), "generation": a.gens, "full_ms": round(full_ms, 2), "ref_pass_ms": round(ref_ms, 2), "env_steps": int(ln.sum()),
           "mean_len": round(float(ln.mean()), 1), "max_len": int(glen.max())}
    cuts = {}
    for name, width in (("le_1499", 1499), ("le_450", 450), ("le_96", 96), ("le_16", 16), ("le_2", 2)):
        T = int((active > width).sum())      # lock-steps with more than `width` pairs active come first (the active count only falls)
        if T == 0:
            cuts[name] = {"lock_steps_before": 0, "ms_before": round(ref_ms, 2)}
            continue
        ms, _ = timed(eng, idx, seeds, T)
        cuts[name] = {"lock_steps_before": T, "ms_before": round(ms, 2)}
    out["cut_at_active_pairs"] = cuts
    prev, seg = ref_ms, {}
    for name, label in (("le_1499", "gt_1499"), ("le_450", "451_1499"), ("le_96", "97_450"), ("le_16", "17_96"), ("le_2", "3_16")):
        seg[label] = round(max(cuts[name]["ms_before"] - prev, 0.0), 2); prev = max(cuts[name]["ms_before"], prev)
    seg["le_2"] = round(full_ms - prev, 2)
    out["ms_by_active_pairs"] = seg
    lo = 0
    for hi in (1, 2, 4, 16, 96, 450, 1499, 10 ** 9):
        out.setdefault("lock_steps_by_active_pairs", {})["%d_%d" % (lo + 1, hi)] = int(((active > lo) & (active <= hi)).sum()); lo = hi
    print(json.dumps(out), flush=True)
    eng.close()
