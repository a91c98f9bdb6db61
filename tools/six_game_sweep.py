#!/usr/bin/env python3
"""Config 5 of BASELINE.json: the 6-game Atari ES sweep (Frostbite / Seaquest / Asteroids / Gravitar / Venture / Zaxxon),
pop 5000 per game, one generation loop per game on this rank's GPU(s).

What differs between the games on the hot path is the action-set size (gym_tensorflow/atari/tf_atari.py:158: Asteroids 14,
the other five 18), which moves every offset behind the output layer in the flat parameter vector.  ALE and the ROMs do not
exist in this image, so every game runs on the SynthAtari fixture with its own action count and its own environment seed
stream -- the numbers measure the engine at that network shape, not the games.

    python tools/six_game_sweep.py --generations 3                  # one GPU, the six games one after the other
    python -m torch.distributed.run --nproc-per-node 8 tools/six_game_sweep.py --generations 3   # population sharded over 8 GPUs
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deep-neuroevolution_amd"))
GAMES = {"frostbite": 18, "seaquest": 18, "asteroids": 14, "gravitar": 18, "venture": 18, "zaxxon": 18}   # tf_atari.py:158


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--generations", type=int, default=3)
    ap.add_argument("--pop", type=int, default=5000)
    ap.add_argument("--tslimit", type=int, default=5000)
    ap.add_argument("--noise-count", type=int, default=250_000_000)
    ap.add_argument("--games", default=",".join(GAMES))
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from dne_hip import _lib, es, policies
    import bench
    cfg = es.Config(**bench.EXP["config"])
    n_pairs = a.pop // 2
    mine = len(es.shard_pairs(n_pairs, rank, world))
    noise = es.SharedNoiseTable(count=a.noise_count)
    for gi, game in enumerate(a.games.split(",")):
        nact = GAMES[game]
        e = _lib.Engine(_lib.KIND_ES, nact, max_members=2 * mine, ref_count=128, device_id=local_rank)
        if world > 1:
            bench.rccl_rendezvous(_lib, e, rank, world)
        noise.attach(e)
        e.set_theta(policies.xavier_flat(nact, seed=gi))
        env = policies.HipAtariEnv(e, seed=1000 * gi)
        ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(gi))) * 255.0).astype(np.uint8)
        e.set_ref_batch(ref)
        e.optimizer_reset()
        e.barrier()
        t0 = time.time(); steps = 0; rets = []
        for gen in range(a.generations):
            rec, ratio = es.es_generation(e, noise.noise.size, cfg, n_pairs, 100 * gi + gen, a.tslimit, bench.EXP["optimizer"], rank, world)
            steps += int(rec["len"].sum()); rets.append(float(rec["ret"].mean()))
        e.barrier()
        wall = time.time() - t0
        if world > 1:
            wall = float(e.comm_allreduce([wall], "max")[0])
        e.check_redzones()
        if rank == 0:
            print(json.dumps({"game": game, "n_actions": nact, "num_params": e.P, "generations": a.generations, "n_gpus": world,
                              "env_steps": steps, "env_steps_per_s": steps / wall, "ms_per_generation": 1e3 * wall / a.generations,
                              "mean_return_by_generation": rets, "data": "SynthAtari fixture with this game's action count"}), flush=True)
        e.close()
        noise._engines.clear()


if __name__ == "__main__":
    main()
