"""TEST-ONLY: the CPU oracle run over the host's usable cores (one forked single-threaded process per core, the big arrays
inherited copy-on-write) for the full-size parity tests.  Only tests may call the oracle; the product never imports this."""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

_BASE = None          # set in the parent right before the fork; read by the worker functions below


def workers():
    from hostinfo import usable_cpus
    return min(usable_cpus(), 256)


def pool_map(fn, items, base):
    global _BASE
    import oracle as O
    O.lib()
    _BASE = base
    try:
        with mp.get_context("fork").Pool(workers()) as pool:
            return pool.map(fn, items, chunksize=1)
    finally:
        _BASE = None


# ---- ES: one antithetic pair with everything the reference's worker reports (es.py:412-439; nses.py:381-384 keeps the RAM trajectory)
def _es_pair(i):
    import oracle as O
    noise, th, ref, idx, seeds, sigma, tslimit, nact, want_bc = _BASE
    L = O.layout(O.KIND_ES, nact)
    out = []
    for s in range(2):
        thp = O.perturb(th, noise, idx[i], sigma, 1 if s == 0 else -1)
        out.append(O.rollout(L, thp, ref, seeds[2 * i + s], tslimit, want_bc=want_bc))
    return out


def es_generation(noise, th, ref, idx, seeds, sigma, tslimit, nact, want_bc=False):
    """all pairs of a generation: returns / sign-returns / lengths [n, 2] (+ the 2n RAM trajectories, member order)"""
    n = len(idx)
    out = pool_map(_es_pair, range(n), (noise, th, ref, idx, seeds, sigma, tslimit, nact, want_bc))
    ret = np.array([[o[0][0], o[1][0]] for o in out], np.float32)
    sg = np.array([[o[0][1], o[1][1]] for o in out], np.float32)
    ln = np.array([[o[0][2], o[1][2]] for o in out], np.int32)
    bcs = [np.array(m[3]) for o in out for m in o] if want_bc else None
    return ret, sg, ln, bcs


# ---- unperturbed rollouts with their RAM trajectory (nses.py:34-39: the archive entries)
def _es_plain(j):
    import oracle as O
    thetas, ref, seeds, tslimit, nact = _BASE
    L = O.layout(O.KIND_ES, nact)
    return np.array(O.rollout(L, thetas[j], ref, seeds[j], tslimit, want_bc=True)[3])


def es_trajectories(thetas, ref, seeds, tslimit, nact):
    return pool_map(_es_plain, range(len(thetas)), (thetas, ref, seeds, tslimit, nact))


# ---- novelty of many trajectories against one archive (nses.py:22-32)
def _novelty(j):
    import oracle as O
    archive, bcs, k = _BASE
    return O.novelty(archive, bcs[j], k)


def novelty_all(archive, bcs, k):
    return np.array(pool_map(_novelty, range(len(bcs)), (archive, bcs, k)), np.float64)


# ---- GA children of cached parents: theta = parent + fl(sigma * noise[fresh]) (ga.py:256-264, the last step of the chain)
def _ga_child(i):
    import oracle as O
    kind, nact, noise, parent_theta, parent, fresh, power, seeds, tslimit = _BASE
    L = O.layout(kind, nact)
    th = O.perturb(parent_theta[parent[i]], noise, int(fresh[i]), float(power[i]), 1)
    return O.rollout(L, th, None, seeds[i], tslimit)[:3]


def ga_children(kind, nact, noise, parent_theta, parent, fresh, power, seeds, tslimit):
    out = pool_map(_ga_child, range(len(parent)), (kind, nact, noise, parent_theta, parent, fresh, power, seeds, tslimit))
    return (np.array([o[0] for o in out], np.float32), np.array([o[1] for o in out], np.float32),
            np.array([o[2] for o in out], np.int32))
