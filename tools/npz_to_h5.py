#!/usr/bin/env python3
"""Turn a dne_hip '.npz' snapshot into the reference's '.h5' (es_distributed/policies.py:49-57: a dataset per TF variable
name, attrs 'name' and 'args_and_kwargs') so that scripts/viz.py:35-36 (`ESAtariPolicy.Load(policy_file)`) and
`initialize_from` (policies.py:345-372) of a stock checkout read it.  Uses h5py, or libhdf5 through dne_hip/h5lite.py.
Only needed for snapshots taken on a machine with neither (the drivers write '.h5' directly when they can).

    python tools/npz_to_h5.py snapshot_iter00020_rew310.npz [out.h5]

args_and_kwargs: the pickle gym 0.9.4's spaces give (policies._dumps_spaces)."""
import os
import pickle
import sys

import numpy as np


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deep-neuroevolution_amd"))
    from dne_hip import h5lite, policies
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else src[:-4] + ".h5"
    with np.load(src, allow_pickle=False) as f:
        (ob_shape, nact), kwargs = pickle.loads(f["__args_and_kwargs__"].tobytes())
        names = [str(n) for n in f["__variables__"]]
        arrays = {n: f["var%03d" % i] for i, n in enumerate(names)}
        cls = str(f["__name__"])
    blob = policies._dumps_spaces(tuple(ob_shape), int(nact), kwargs)
    try:
        import h5py
    except ImportError:
        h5lite.write_snapshot(dst, arrays, cls, blob)
    else:
        with h5py.File(dst, "w", libver="latest") as f:
            for k, v in arrays.items():
                f[k] = v
            f.attrs["name"] = cls
            f.attrs["args_and_kwargs"] = np.void(blob)
    print("wrote", dst, "with", len(arrays), "variables")


if __name__ == "__main__":
    main()
