#!/usr/bin/env python3
"""Config 5 of BASELINE.json: the 6-game Atari ES loop (Frostbite / Seaquest / Asteroids / Gravitar / Venture / Zaxxon), pop 5000
per game, on one GPU (tools/workloads.py:six_games).  What differs between the games on the hot path is the action-set size
(gym_tensorflow/atari/tf_atari.py:158: Asteroids 14, the other five 18), which moves every offset behind the output layer in the
flat parameter vector; ALE and the ROMs do not exist in this image, so every game runs on the SynthAtari fixture with its own
action count and seed streams.  On N GPUs:  python bench.py --gpus N --extra sweep  (the population sharded over the ranks)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as W
from dne_hip import es

ap = argparse.ArgumentParser()
ap.add_argument("--generations", type=int, default=3); ap.add_argument("--pop", type=int, default=5000)
ap.add_argument("--tslimit", type=int, default=5000); ap.add_argument("--games", default=",".join(W.GAMES))
a = ap.parse_args()
r = W.six_games(es.SharedNoiseTable(), generations=a.generations, pop=a.pop, tslimit=a.tslimit, games=a.games.split(","))
for g in r.pop("games"):
    print(json.dumps(g))
print(json.dumps(r))
