#!/usr/bin/env python3
"""Config 3: Deep GA, 1000 children per generation, top-20 parents, sigma 0.005 -- evaluation
throughput of generation 0 (every child its own normc genome) and of later generations (children of cached parents)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
from dne_hip import _lib, es

n, T, sigma, tslimit = 1000, 20, 0.005, 5000
LARGE = "--large" in sys.argv     # the GPU tree's protocol: LargeModel, genomes ((idx0,), (idx, power), ...), mutation power 0.002
if LARGE:
    from dne_hip import ga_gpu
    sigma = 0.002                 # configurations/ga_atari_config.json
e = _lib.Engine(_lib.KIND_GA_LARGE if LARGE else _lib.KIND_GA, 18, max_members=n, profile_events=True)
noise = es.SharedNoiseTable(); noise.attach(e)
if LARGE:
    e.ga_set_init_scale(ga_gpu.model_scale_by(18, _lib.KIND_GA_LARGE))
rs = np.random.RandomState(0)
pop, score = [], np.array([], np.float32)
for gen in range(3 if LARGE else 4):
    if LARGE:
        chains = [(tuple(pop[rs.randint(len(pop))]) + ((int(noise.sample_index(rs, e.P)), sigma),)) if pop else (int(noise.sample_index(rs, e.P)),) for _ in range(n)]
    else:
        chains = [(list(pop[rs.randint(len(pop))]) if pop else []) + [int(noise.sample_index(rs, e.P))] for _ in range(n)]
    seeds = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    t0 = time.time()
    ret, sg, ln = e.ga_eval_powers(chains, tslimit, seeds) if LARGE else e.ga_eval(chains, sigma, tslimit, seeds)
    wall = time.time() - t0
    allc = [c for c in pop[:1]] + list(chains); allr = np.concatenate([score[:1], ret]).astype(np.float32)
    sel = e.ga_select(allr, T); pop = [allc[i] for i in sel]; score = allr[sel]
    p = e.profile()
    print(json.dumps({"gen": gen, "wall_s": round(wall, 3), "env_steps": int(ln.sum()), "steps_per_s": round(ln.sum() / wall),
                      "mean_len": float(ln.mean()), "best": float(score[0]), "fc_ms": round(p["fc_ms"], 1), "conv_ms": round(p["conv_ms"], 1),
                      "env_ms": round(p["env_ms"], 1), "eval_ms": round(p["eval_ms"], 1)}))
