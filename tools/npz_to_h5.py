#!/usr/bin/env python3
"""Turn a dne_hip '.npz' snapshot into the reference's '.h5' (es_distributed/policies.py:49-57: a dataset per TF variable
name, attrs 'name' and 'args_and_kwargs') so that scripts/viz.py:35-36 (`ESAtariPolicy.Load(policy_file)`) and
`initialize_from` (policies.py:345-372) of a stock checkout read it.  Needs h5py (absent from the build image).

    python tools/npz_to_h5.py snapshot_iter00020_rew310.npz [out.h5]

args_and_kwargs: the reference pickles the gym spaces it was constructed with; gym is needed to build them, so this
script does so when gym is importable and otherwise stores (shape, n_actions) -- enough for dne_hip's own Load."""
import pickle
import sys

import numpy as np


def main():
    import h5py
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else src[:-4] + ".h5"
    with np.load(src, allow_pickle=False) as f:
        (ob_shape, nact), kwargs = pickle.loads(f["__args_and_kwargs__"].tobytes())
        names = [str(n) for n in f["__variables__"]]
        arrays = {n: f["var%03d" % i] for i, n in enumerate(names)}
        cls = str(f["__name__"])
    try:
        import gym
        args = (gym.spaces.Box(low=0.0, high=1.0, shape=tuple(ob_shape)), gym.spaces.Discrete(int(nact)))
    except Exception:
        args = (tuple(ob_shape), int(nact))
    with h5py.File(dst, "w", libver="latest") as f:
        for k, v in arrays.items():
            f[k] = v
        f.attrs["name"] = cls
        f.attrs["args_and_kwargs"] = np.void(pickle.dumps((args, kwargs), protocol=-1))
    print("wrote", dst, "with", len(arrays), "variables")


if __name__ == "__main__":
    main()
