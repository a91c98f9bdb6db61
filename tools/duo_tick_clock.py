#!/usr/bin/env python3
"""What a tick of k_fc_duo's table timeline costs, by how many of the workgroup's eight units are streaming in it.
Profiling build only (make -C deep-neuroevolution_amd/csrc clock -> libdne_hip_clock.so): for 64 workgroups spread over the launch,
lane 0 of every wave stamps the shader clock in front of and behind the s_barrier of every tick of the workgroup's SECOND work item
(steady state: the chip is full, the first items' start-up is over); the item's plan (every wave's start delay and length in ticks)
comes with it, so the host knows which units were active in which tick.
    DNE_LIB_PATH=.../libdne_hip_clock.so DNE_NSUB=1 python tools/duo_tick_clock.py [--pairs 2500]
Output (JSON): per active-unit count the mean tick duration; for an active wave its own time inside a tick (rows: s_waitcnt + math +
refill) against the time it then waits at the barrier; the same split for the wave that leads the timeline (the first toucher of a
table row) against the followers; the item as a whole."""
import argparse, ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=2500)
ap.add_argument("--tslimit", type=int, default=4)
a = ap.parse_args()
NBLK = 121
e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * a.pairs, ref_count=128, profile_events=True)
noise = es.SharedNoiseTable(count=250_000_000); noise.attach(e)
e.set_theta(policies.xavier_flat(18, 0))
env = policies.HipAtariEnv(e, seed=0)
ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
e.set_ref_batch(ref)
_, idx, seeds = es.generation_inputs(noise.noise.size, e.P, a.pairs, 0, 0, 1)
e.es_eval(idx, 0.02, a.tslimit, seeds)
p = e.profile()
WGS, TMAX = 64, 288
NW = 8   # k_fc_duo fills waves 0-3 (two units each), k_fc_ring all eight (one unit each)
ticks = np.zeros((WGS, NW, TMAX, 2), np.int64)
plan = np.zeros((WGS, NW, 8), np.int64)
fn = e.lib.dne_debug_duo_ticks
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = C.c_int
if fn(e.h, ticks.ctypes.data_as(C.c_void_p), plan.ctypes.data_as(C.c_void_p)) != 0:
    raise SystemExit("this library has no tick clock: build it with make clock and set DNE_LIB_PATH")

by_units = {}      # active units -> [tick durations in shader cycles]
work = {1: [], 2: []}      # an active wave's own time inside a tick, by its active units
wait = {1: [], 2: []}      # ... and its wait at the barrier
lead_work, lead_wait, foll_work, foll_wait = [], [], [], []
idle_wait = []
items = []
for g in range(WGS):
    pl = plan[g]
    n = int(pl[:, 3].max())
    if n <= 1 or n > TMAX:
        continue
    tmax = int(pl[0, 2])
    mhz = [(pl[w, 5] - pl[w, 4]) / max((pl[w, 7] - pl[w, 6]) * 0.01, 1e-9) for w in range(NW) if pl[w, 7] > pl[w, 6]]
    waves = [w for w in range(NW) if pl[w, 3] > 0]
    act = np.zeros((NW, n), np.int32)   # active units of wave w in tick i
    for w in waves:
        d, ln = int(pl[w, 0]), int(pl[w, 1])
        if ln == 0:
            continue
        gb = ln - NBLK if ln > NBLK else 0
        for i in range(d, min(d + ln, n)):
            k = i - d
            a_on = k < NBLK
            b_on = ln > NBLK and gb <= k < gb + NBLK
            act[w, i] = int(a_on) + int(b_on)
    after = ticks[g, :, :n, 1].astype(np.float64)
    before = ticks[g, :, :n, 0].astype(np.float64)
    t_end = after[waves].max(axis=0)             # the barrier releases every wave at (nearly) the same time
    for i in range(1, n):
        dur = t_end[i] - t_end[i - 1]
        by_units.setdefault(int(act[:, i].sum()), []).append(dur)
        first = min((w for w in waves if act[w, i] > 0), default=-1)
        for w in waves:
            wk, wt = before[w, i] - after[w, i - 1], after[w, i] - before[w, i]
            if act[w, i] > 0:
                work[int(act[w, i])].append(wk); wait[int(act[w, i])].append(wt)
                if w == first:
                    lead_work.append(wk); lead_wait.append(wt)
                else:
                    foll_work.append(wk); foll_wait.append(wt)
            else:
                idle_wait.append(wt)
    items.append({"ticks": n, "tmax": tmax, "unit_ticks": int(act.sum()), "shader_mhz": round(float(np.mean(mhz)), 1) if mhz else None,
                  "item_us": round(float((pl[:, 7].max() - pl[:, 6].min()) * 0.01), 1)})
mhz = float(np.mean([it["shader_mhz"] for it in items if it["shader_mhz"]])) if items else 0.0
us = lambda cyc: round(float(np.mean(cyc)) / mhz, 3) if len(cyc) and mhz else None
out = {"pairs": a.pairs, "fc_ms_per_launch": p["fc_ms"] / max(p["fc_launches"], 1), "sampled_items": len(items), "shader_mhz": round(mhz, 1),
       "item": {"ticks_mean": float(np.mean([it["ticks"] for it in items])), "unit_ticks_mean": float(np.mean([it["unit_ticks"] for it in items])),
                "us_mean": float(np.mean([it["item_us"] for it in items])),
                "fill": float(np.mean([it["unit_ticks"] / (8.0 * it["ticks"]) for it in items])) if items else None},
       "tick_us_by_active_units": {str(k): {"n": len(v), "us": us(v)} for k, v in sorted(by_units.items())},
       "active_wave_by_its_units": {str(k): {"own_us": us(work[k]), "barrier_wait_us": us(wait[k])} for k in (1, 2)},
       "first_wave_of_the_timeline": {"own_us": us(lead_work), "barrier_wait_us": us(lead_wait)},
       "other_active_waves": {"own_us": us(foll_work), "barrier_wait_us": us(foll_wait)},
       "idle_wave_barrier_wait_us": us(idle_wait)}
print(json.dumps(out, indent=1))
