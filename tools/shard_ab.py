#!/usr/bin/env python3
"""An N-rank launch rehearsed on one GPU (tools/workloads.py:simulate_ranks) under different index-sharding modes and engine knobs:
    python tools/shard_ab.py --worlds 2,4,8 --gens 6 --warmup 3 "uniform" "table" "table DNE_FC_RING=2 DNE_RING_MIN=0"
Each setting = a shard mode followed by DNE_* knobs (read at dne_create).  One JSON line per (world, setting)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from dne_hip import es
import workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--gens", type=int, default=6)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--verify", type=int, default=0)
a = ap.parse_args()
noise = es.SharedNoiseTable()
for w in [int(x) for x in a.worlds.split(",")]:
    for st in a.settings:
        parts = st.split()
        env = dict(kv.split("=") for kv in parts[1:])
        env = {k: v.replace("{share}", str(len(es.shard_pairs(2500, 0, w)))).replace("{share60}", str(int(0.6 * len(es.shard_pairs(2500, 0, w))))) for k, v in env.items()}
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            r = W.simulate_ranks(noise, w, steps=a.gens, warmup=a.warmup, verify_generations=a.verify, shard=parts[0])
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        print(json.dumps({"world": w, "setting": st, "knobs": env, "ms_per_generation": round(r["ms_per_step"], 2), "rank0_ms": r["rank0_share_ms_per_generation"],
                          "value": round(r["value"]), "theta_sha": r["theta_sha256"][:16]}), flush=True)
