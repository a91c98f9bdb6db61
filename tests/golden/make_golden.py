#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference code (read-only tree at /root/reference).

Runs only in the build container (the reference tree does not exist on the GPU box); its output
tests/golden/reference_vectors.npz is committed.  What is imported from the reference, unmodified:

  es_distributed/es.py            compute_ranks, compute_centered_ranks, batched_weighted_sum, itergroups,
                                  SharedNoiseTable.sample_index/get            (redis stubbed: transport only)
  es_distributed/optimizers.py    Adam, SGD
  es_distributed/atari_wrappers.py  wrap_deepmind (NoopReset/MaxAndSkip/FireReset/WarpFrame/FrameStack/
                                  ScaledFloat) over the SynthAtari fixture, with the real PIL  (gym stubbed
                                  with the gym-0.9.4 Wrapper contract: step->_step, reset->_reset)
  es_distributed/nses.py          euclidean_distance, compute_novelty_vs_archive (tensorflow stubbed; np.float
                                  alias restored because numpy>=1.24 removed it)

Third-party modules that are absent here (tensorflow, gym, ALE, redis) are stubbed ONLY so that the
reference modules import; no arithmetic comes from a stub.

Usage:  python tests/golden/make_golden.py [--full-noise]
  --full-noise additionally builds the real 250M-entry noise table (minutes, 3 GB) and records its
  head/tail/checksum.
"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, REF)

# ---------------------------------------------------------------- stubs (import-only)
sys.modules["redis"] = types.ModuleType("redis")
tf = types.ModuleType("tensorflow")
sys.modules["tensorflow"] = tf

gym = types.ModuleType("gym")


class _Env:
    def step(self, action):
        return self._step(action)

    def reset(self):
        return self._reset()

    @property
    def unwrapped(self):
        return self


class _Wrapper(_Env):  # gym 0.9.4 core.Wrapper contract
    def __init__(self, env):
        self.env = env
        self.action_space = getattr(env, "action_space", None)
        self.observation_space = getattr(env, "observation_space", None)

    def _step(self, action):
        return self.env.step(action)

    def _reset(self):
        return self.env.reset()

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def spec(self):
        return self.env.spec


class _ObservationWrapper(_Wrapper):
    def _reset(self):
        return self._observation(self.env.reset())

    def _step(self, action):
        ob, r, d, info = self.env.step(action)
        return self._observation(ob), r, d, info


class _Box:
    def __init__(self, low, high, shape=None):
        self.low, self.high, self.shape = low, high, shape


class _Discrete:
    def __init__(self, n):
        self.n = n


gym.Env, gym.Wrapper, gym.ObservationWrapper = _Env, _Wrapper, _ObservationWrapper
spaces = types.ModuleType("gym.spaces")
spaces.Box, spaces.Discrete = _Box, _Discrete
gym.spaces = spaces
sys.modules["gym"] = gym
sys.modules["gym.spaces"] = spaces
if not hasattr(np, "float"):
    np.float = float  # nses.py:24 uses the alias removed in numpy 1.24

from es_distributed import es as ref_es  # noqa: E402
from es_distributed import optimizers as ref_opt  # noqa: E402
from es_distributed import atari_wrappers as ref_wrap  # noqa: E402
from es_distributed import nses as ref_nses  # noqa: E402

import oracle as O  # noqa: E402  (fixture env only: SynthAtari raw frames + palette)

ACTION_MEANINGS = ["NOOP", "FIRE", "UP", "RIGHT", "LEFT", "DOWN", "UPRIGHT", "UPLEFT", "DOWNRIGHT", "DOWNLEFT",
                   "UPFIRE", "RIGHTFIRE", "LEFTFIRE", "DOWNFIRE", "UPRIGHTFIRE", "UPLEFTFIRE", "DOWNRIGHTFIRE",
                   "DOWNLEFTFIRE"]


class SynthAtariGym(_Env):
    """gym-shaped adapter over the SynthAtari raw frames (frameskip 1, RGB observations)."""

    class _Spec:
        id = "FrostbiteNoFrameskip-v4"

    class _NoopRng:
        def __init__(self, outer):
            self.outer = outer

        def randint(self, lo, hi):
            assert (lo, hi) == (1, 31)
            return 1 + self.outer.seed_value % 30  # noop count fixed by the episode seed

    def __init__(self):
        self.spec = self._Spec()
        self.action_space = _Discrete(18)
        self.observation_space = _Box(0, 255, (210, 160, 3))
        self.np_random = self._NoopRng(self)
        self.seed_value = 0
        self.pal = O.palette()
        self.ram = None

    def get_action_meanings(self):
        return ACTION_MEANINGS

    def _obs(self):
        return self.pal[O.raw_render(self.ram)]

    def _reset(self):
        self.ram = O.raw_reset(self.seed_value)
        return self._obs()

    def _step(self, a):
        rew = O.raw_frame(self.ram, a)
        return self._obs(), float(rew), bool(self.ram[9]), {}


def golden_wrappers(out):
    seeds = [0, 7, 29, 123456789]
    rs = np.random.RandomState(42)
    for s in seeds:
        raw = SynthAtariGym()
        raw.seed_value = s
        env = ref_wrap.wrap_deepmind(raw)
        ob = env.reset()
        assert ob.dtype == np.float32 and ob.shape == (84, 84, 4)
        u8 = np.rint(ob * 255.0).astype(np.uint8)
        assert np.array_equal(u8.astype(np.float32) / 255.0, ob)  # ScaledFloatFrame is exactly u8/255
        actions = rs.randint(0, 18, size=24)
        actions[:4] = [5, 5, 3, 13]  # make sure something happens (DOWN, DOWN, RIGHT, DOWNFIRE)
        obs, rews, dones, rams = [u8], [], [], []
        for a in actions:
            ob, r, d, _ = env.step(int(a))
            obs.append(np.rint(ob * 255.0).astype(np.uint8))
            rews.append(r)
            dones.append(d)
            rams.append(raw.ram.copy())
            if d:
                break
        out["wrap_s%d_actions" % s] = actions[:len(rews)].astype(np.int32)
        out["wrap_s%d_obs" % s] = np.stack(obs)
        out["wrap_s%d_rews" % s] = np.array(rews, np.float32)
        out["wrap_s%d_dones" % s] = np.array(dones, np.bool_)
        out["wrap_s%d_rams" % s] = np.stack(rams)
    out["wrap_seeds"] = np.array(seeds, np.int64)
    # WarpFrame alone on the SURVEY's random RGB frame
    rgb = np.random.RandomState(1).randint(0, 256, (210, 160, 3)).astype(np.uint8)
    wf = ref_wrap.WarpFrame(SynthAtariGym())
    out["warp_rs1"] = wf._observation(rgb)[:, :, 0]
    # an episode driven to game over by DOWN presses (early-done inside a skipped step)
    raw = SynthAtariGym()
    raw.seed_value = 5
    env = ref_wrap.wrap_deepmind(raw)
    env.reset()
    obs, rews = [], []
    for t in range(400):
        ob, r, d, _ = env.step(5)
        obs.append(np.rint(ob * 255.0).astype(np.uint8))
        rews.append(r)
        if d:
            break
    assert d, "episode should end under constant DOWN"
    out["wrap_down_len"] = np.array(len(rews))
    out["wrap_down_rews"] = np.array(rews, np.float32)
    out["wrap_down_last_obs"] = obs[-1]
    out["wrap_down_obs_sha"] = np.frombuffer(hashlib.sha256(np.stack(obs).tobytes()).digest(), np.uint8)


def golden_reduce(out):
    # es.py:70-85
    x = np.array([[10, 0], [0, 30], [20, 20]], np.float32)
    out["ranks_small_in"] = x
    out["ranks_small_out"] = ref_es.compute_centered_ranks(x)
    rs = np.random.RandomState(1)
    # distinct returns: argsort tie order is irrelevant -> pins the formula exactly
    xd = rs.permutation(5000).astype(np.float32).reshape(2500, 2) * 10
    out["ranks_distinct_in"] = xd
    out["ranks_distinct_out"] = ref_es.compute_centered_ranks(xd)
    # Frostbite-like tied returns; reference tie order is implementation-defined
    xt = (10 * rs.poisson(20, (128, 2))).astype(np.float32)
    out["ranks_tied_in"] = xt
    out["ranks_tied_out"] = ref_es.compute_centered_ranks(xt)

    # es.py:115-122 batched_weighted_sum over a small table with the reference's own get()
    table = ref_es.SharedNoiseTable.__new__(ref_es.SharedNoiseTable)
    table.noise = np.random.RandomState(123).randn(2_000_000).astype(np.float32)
    out["noise_small_head"] = table.noise[:8].copy()
    P = 10007
    srs = np.random.RandomState(0)
    idx = np.array([table.sample_index(srs, P) for _ in range(300)])
    out["ws_idx"] = idx.astype(np.int64)
    proc = ref_es.compute_centered_ranks(xt[:150].copy().repeat(2, axis=0)[:300])
    w = proc[:, 0] - proc[:, 1]
    w = rs.randn(300).astype(np.float32)
    out["ws_w"] = w
    g, count = ref_es.batched_weighted_sum(w, (table.get(i, P) for i in idx), batch_size=500)
    assert count == 300
    out["ws_g"] = g.astype(np.float32)
    g7, _ = ref_es.batched_weighted_sum(w, (table.get(i, P) for i in idx), batch_size=7)
    out["ws_g_b7"] = g7.astype(np.float32)
    out["ws_P"] = np.array(P)

    # sample_index with the real table length (es.py:66-67): randint(0, 250e6 - dim + 1)
    class _Len:
        def __len__(self):
            return 250_000_000
    t2 = ref_es.SharedNoiseTable.__new__(ref_es.SharedNoiseTable)
    t2.noise = _Len()
    for dim in (1009058, 1008450):
        srs = np.random.RandomState(0)
        out["sample_index_%d" % dim] = np.array([t2.sample_index(srs, dim) for _ in range(8)], np.int64)


def golden_optimizers(out):
    rs = np.random.RandomState(3)
    P = 4096
    theta0 = (rs.randn(P) * 0.05).astype(np.float32)
    gs = [(rs.randn(P) * 1e-3).astype(np.float32) for _ in range(4)]
    out["opt_theta0"] = theta0
    out["opt_gs"] = np.stack(gs)
    l2 = 0.005
    for name, mk in (("adam", lambda th: ref_opt.Adam(th, stepsize=0.01)),
                     ("sgd", lambda th: ref_opt.SGD(th, stepsize=0.01, momentum=0.9))):
        opt = mk(theta0.copy())
        thetas, ratios = [], []
        theta = theta0.copy()
        for g in gs:
            ratio, theta = opt.update(-g + l2 * theta)  # es.py:298
            thetas.append(np.asarray(theta, np.float64))
            ratios.append(float(ratio))
        out["opt_%s_thetas" % name] = np.stack(thetas)  # float64 under numpy>=2 (NEP 50)
        out["opt_%s_ratios" % name] = np.array(ratios)
        out["opt_%s_dtype" % name] = np.array(str(np.asarray(theta).dtype))


def golden_novelty(out):
    rs = np.random.RandomState(4)
    lens = [37, 50, 12, 50, 80, 5, 64, 33, 41, 50, 9, 77]
    arch = [rs.randint(0, 256, (n, 128)).astype(np.uint8) for n in lens]
    bc = rs.randint(0, 256, (50, 128)).astype(np.uint8)
    out["nov_lens"] = np.array(lens)
    out["nov_arch"] = np.concatenate(arch)
    out["nov_bc"] = bc
    out["nov_dists"] = np.array([ref_nses.euclidean_distance(p.astype(float), bc.astype(float)) for p in arch])
    out["nov_k10"] = np.array(ref_nses.compute_novelty_vs_archive(arch, bc, 10))
    out["nov_k3"] = np.array(ref_nses.compute_novelty_vs_archive(arch, bc, 3))


def golden_normc(out):
    # tf_util.py:122-130 _normalize body, verbatim arithmetic on a numpy array (the function itself
    # is a nested closure inside a TF py_func and cannot be called without TensorFlow)
    rs = np.random.RandomState(5)
    for name, shape, std in (("a", (8, 8, 4, 16), 1.0), ("b", (3872, 8), 1.0), ("c", (256, 18), 0.1)):
        w = rs.randn(*shape).astype(np.float32)
        o = np.reshape(w.copy(), [-1, shape[-1]])
        o *= std / np.sqrt(np.square(o).sum(axis=0, keepdims=True))
        out["normc_%s_in" % name] = w
        out["normc_%s_out" % name] = np.reshape(o, shape)


def golden_full_noise(out):
    # es.py:51-61 exactly (without the multiprocessing.Array backing store)
    noise = np.random.RandomState(123).randn(250_000_000).astype(np.float32)
    out["noise_full_head"] = noise[:8].copy()
    out["noise_full_tail"] = noise[-3:].copy()
    out["noise_full_sha256"] = np.frombuffer(hashlib.sha256(noise.tobytes()).digest(), np.uint8)
    out["noise_full_sum64"] = np.array(noise.astype(np.float64).sum())


def main():
    out = {}
    golden_reduce(out)
    golden_optimizers(out)
    golden_novelty(out)
    golden_normc(out)
    golden_wrappers(out)
    path = os.path.join(HERE, "reference_vectors.npz")
    if "--full-noise" in sys.argv:
        golden_full_noise(out)
    elif os.path.exists(path):  # keep previously recorded full-table values
        old = np.load(path)
        for k in old.files:
            if k.startswith("noise_full_"):
                out[k] = old[k]
    np.savez_compressed(path, **out)
    print("wrote", path, "keys:", len(out), "bytes:", os.path.getsize(path))


if __name__ == "__main__":
    main()
