#!/usr/bin/env python3
"""Per-WORKGROUP clock of the lock-step's kernels (profiling build: make -C deep-neuroevolution_amd/csrc clock; DNE_LIB_PATH=.../libdne_hip_clock.so).
rocprofv3 --pmc serialises the dispatches, so counters cannot say why a kernel that takes X alone takes 4X beside the streaming fc; this can:
every workgroup of k_conv12 / k_fc_ring / k_out / k_env_logic / k_env_render leaves {kernel, block, start, end (100 MHz wall clock), HW_ID, XCC_ID}.
Per setting (knobs as in tools/ab_inproc.py, e.g. "DNE_NSUB=1" = one window: every kernel alone) and per kernel:
  workgroup duration (mean / median / p90), workgroups resident per CU while the kernel runs, and -- for kernels other than the streaming fc --
  the same split by whether a streaming-fc workgroup was resident on the SAME CU for most of the workgroup's life.
    DNE_LIB_PATH=.../libdne_hip_clock.so python tools/wg_clock.py "DNE_NSUB=1" "X=0" [--pairs 2500] [--steps 6]"""
import argparse, ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es, policies

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--pairs", type=int, default=2500)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--noise-count", type=int, default=250_000_000)
a = ap.parse_args()
NAMES = {0: "k_conv12", 1: "k_fc_ring", 2: "k_out", 3: "k_env_logic", 4: "k_env_render", 5: "k_fc_duo", 6: "k_fc_sub", 7: "k_tail_step"}
STREAM = (1, 5, 6)   # the streaming fc kernels
CAP = 1 << 19
noise = es.SharedNoiseTable(count=a.noise_count)
ref = None


def engine(env):
    global ref
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = _lib.Engine(_lib.KIND_ES, 18, max_members=2 * a.pairs, ref_count=128)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    noise.attach(e)
    e.set_theta(policies.xavier_flat(18, 0))
    if ref is None:
        envh = policies.HipAtariEnv(e, seed=0)
        ref = np.rint(np.stack(es.get_ref_batch(envh, 128, np.random.RandomState(0))) * 255.0).astype(np.uint8)
    e.set_ref_batch(ref)
    return e


def records(e, reset):
    fn = e.lib.dne_debug_wg_clock
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_uint), C.c_int]; fn.restype = C.c_int
    buf = np.zeros((CAP, 4), np.int64); n = C.c_uint(0)
    if fn(e.h, buf.ctypes.data_as(C.c_void_p), CAP, C.byref(n), 1 if reset else 0) != 0:
        raise SystemExit("this library has no workgroup clock: build it with make clock and set DNE_LIB_PATH")
    return buf[:min(n.value, CAP)], n.value


def overlap_frac(s0, s1, iv):
    """fraction of [s0, s1) covered by the (sorted, possibly overlapping) intervals iv = [(b, e)]"""
    cov, hi = 0, s0
    for b, e in iv:
        if e <= hi or b >= s1:
            continue
        lo = max(b, hi)
        cov += min(e, s1) - lo
        hi = min(e, s1)
    return cov / max(s1 - s0, 1)


out = []
for s in a.settings:
    env = dict(kv.split("=", 1) for kv in s.split() if "=" in kv)
    e = engine(env)
    _, idx, seeds = es.generation_inputs(noise.noise.size, e.P, a.pairs, 0, 0, 1)
    e.es_eval(idx, 0.02, a.steps, seeds)
    records(e, True)
    e.es_eval(idx, 0.02, a.steps, seeds)
    rec, total = records(e, True)
    e.close(); noise._engines[:] = []
    kid = (rec[:, 0] >> 32).astype(int); t0 = rec[:, 1]; t1 = rec[:, 2]
    hw = rec[:, 3] & 0xffffffff; xcc = (rec[:, 3] >> 32) & 0xf
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)   # (XCC, SE, SH, CU)
    res = {"setting": s, "pairs": a.pairs, "lock_steps": a.steps, "records": int(total), "kept": int(len(rec)), "distinct_cus": int(len(set(cu.tolist()))), "kernels": {}}
    # streaming-fc residency per CU
    stream_iv = {}
    for i in np.nonzero(np.isin(kid, STREAM))[0]:
        stream_iv.setdefault(int(cu[i]), []).append((int(t0[i]), int(t1[i])))
    for v in stream_iv.values():
        v.sort()
    for k in sorted(set(kid.tolist())):
        m = kid == k
        d = (t1[m] - t0[m]) * 0.01   # us
        span = (t1[m].max() - t0[m].min()) * 0.01
        # time the kernel has at least one workgroup in flight (union of its workgroups' lives), and workgroup-time per CU over that
        order = np.argsort(t0[m]); b = t0[m][order]; en = t1[m][order]
        uni, hi = 0, -1
        for bb, ee in zip(b.tolist(), en.tolist()):
            if bb > hi: uni += ee - bb; hi = ee
            elif ee > hi: uni += ee - hi; hi = ee
        uni *= 0.01
        r = {"workgroups": int(m.sum()), "wg_us_mean": round(float(d.mean()), 2), "wg_us_median": round(float(np.median(d)), 2), "wg_us_p90": round(float(np.percentile(d, 90)), 2),
             "busy_us_total": round(uni, 1), "resident_wgs_per_cu_while_running": round(float(d.sum()) / max(uni, 1e-9) / max(res["distinct_cus"], 1), 2)}
        if k not in STREAM and stream_iv:
            fr = np.array([overlap_frac(int(x0), int(x1), stream_iv.get(int(c), [])) for x0, x1, c in zip(t0[m], t1[m], cu[m])])
            with_s, without = d[fr >= 0.8], d[fr <= 0.2]
            r["beside_streaming_fc"] = {"workgroups": int(len(with_s)), "wg_us_mean": round(float(with_s.mean()), 2) if len(with_s) else None}
            r["cu_without_streaming_fc"] = {"workgroups": int(len(without)), "wg_us_mean": round(float(without.mean()), 2) if len(without) else None}
        res["kernels"][NAMES.get(k, str(k))] = r
    print(json.dumps(res), flush=True)
    out.append(res)
