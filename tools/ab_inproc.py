#!/usr/bin/env python3
"""Same-box, same-process A/B of engine knobs: one noise table, one engine per setting (the DNE_* knobs are read at dne_create),
settings run round-robin `--rounds` times so that drift shows.  Per setting:
  alone_fc_ms    the streaming fc kernel alone: 2500 pairs in ONE window, nobody dies for 6 lock-steps (HIP events per launch)
  lockstep_ms    a full-width lock-step of the default window count (wall / steps, reference pass subtracted)
  gen_ms         ms per generation of the driver's workload (generations g0 .. g0+n-1 after `--warmup`, theta evolving)
    python tools/ab_inproc.py "X=0" "DNE_DUO_SYNC=2" "DNE_DUO_SYNC=2 DNE_NSUB_FULL=3"
"""
import argparse, hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--pairs", type=int, default=2500)
ap.add_argument("--gens", type=int, default=8)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--skip", default="", help="comma list of alone,lockstep,gen to leave out")
ap.add_argument("--idx-align", type=int, default=1, help="timing experiment (fixed-width legs only): noise indices rounded down to a multiple of this")
a = ap.parse_args()
skip = set(a.skip.split(","))
noise = es.SharedNoiseTable()
CFG = es.Config(l2coeff=0.005, noise_stdev=0.02, episodes_per_batch=2 * a.pairs, timesteps_per_batch=10000, calc_obstat_prob=0.0,
                eval_prob=0.0, snapshot_freq=0, return_proc_mode="centered_rank", episode_cutoff_mode=5000)
OPT = {"type": "adam", "args": {"stepsize": 0.01}}
ref = None


def engine(env):
    global ref
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * a.pairs, ref_count=128, profile_events=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    noise.attach(e)
    e.set_theta(policies.xavier_flat(18, 0))
    if ref is None:
        envh = policies.HipAtariEnv(e, seed=0)
        ref = np.rint(np.stack(es.get_ref_batch(envh, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
    e.set_ref_batch(ref)
    e.optimizer_reset()
    return e


def fixed_width(e, T=6):
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, a.pairs, 0, 0, 1)
    idx = idx - idx % a.idx_align
    e.es_eval(idx, 0.02, T, seeds)
    t = time.time(); e.es_eval(idx, 0.02, T, seeds); wall = time.time() - t
    p = e.profile()
    return {"fc_ms": p["fc_ms"] / max(p["fc_launches"], 1), "conv_ms": p["conv_ms"] / max(p["fc_launches"], 1),
            "step_wall_ms": (wall * 1e3 - p["ref_ms"]) / T, "launches": p["fc_launches"]}


res = {}
for rnd in range(a.rounds):
    for s in a.settings:
        env = dict(kv.split("=", 1) for kv in s.split() if "=" in kv)
        r = res.setdefault(s, {"alone_fc_ms": [], "lockstep_ms": [], "gen_ms": [], "steps_per_s": []})
        if "alone" not in skip:
            e = engine(dict(env, DNE_NSUB="1"))
            r["alone_fc_ms"].append(round(fixed_width(e)["fc_ms"], 4))
            e.close()
        e = engine(env)
        if "lockstep" not in skip:
            fw = fixed_width(e)
            r["lockstep_ms"].append(round(fw["step_wall_ms"], 4))
            r.setdefault("lockstep_fc_ms_per_launch", []).append(round(fw["fc_ms"], 4)); r.setdefault("lockstep_conv_ms_per_launch", []).append(round(fw["conv_ms"], 4))
        if "gen" not in skip:
            e.set_theta(policies.xavier_flat(18, 0)); e.optimizer_reset()
            for g in range(a.warmup):
                es.es_generation(e, noise.noise.size, CFG, a.pairs, g, 5000, OPT)
            e.barrier(); t = time.time(); steps = 0
            for g in range(a.warmup, a.warmup + a.gens):
                rec, _ = es.es_generation(e, noise.noise.size, CFG, a.pairs, g, 5000, OPT)
                steps += int(rec["len"].sum())
            e.barrier(); wall = time.time() - t
            r["gen_ms"].append(round(1e3 * wall / a.gens, 2)); r["steps_per_s"].append(round(steps / wall))
            r.setdefault("theta_sha", []).append(hashlib.sha256(e.get_theta().tobytes()).hexdigest()[:16])   # settings that only change a schedule end on the same bits
        e.close()
        noise._engines[:] = []
        print(json.dumps({"round": rnd, "setting": s, **{k: v[-1] for k, v in r.items() if v}}), flush=True)
print(json.dumps({"summary": {s: {k: (min(v) if v and k != "theta_sha" else sorted(set(v)) if v else None) for k, v in r.items()} for s, r in res.items()}}))
