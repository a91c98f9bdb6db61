"""ctypes binding of the CPU ORACLE (oracle/libdne_oracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package (deep-neuroevolution_amd/).
See oracle/dne_oracle.h for the reference file:line each function restates and for the
pinning status ("parity unpinned" for the TF forward numerics and the ALE emulator).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdne_oracle.so")

KIND_ES, KIND_GA, KIND_GA_LARGE = 0, 1, 2
OB_SHAPE = (84, 84, 4)
OB_BYTES = 84 * 84 * 4
RAM = 128
BN_FLOATS = 608
ENV_MAX_EPISODE_STEPS = 400000


def build(force=False):
    src = os.path.join(_HERE, "dne_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class Layout(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "kind", "nact", "P", "c1w", "c1b", "bn1b", "bn1g", "c2w", "c2b", "bn2b", "bn2g",
        "fcw", "fcb", "bn3b", "bn3g", "ow", "ob", "c3w", "c3b")]


class WEnv(C.Structure):
    _fields_ = [("ram_prev", C.c_uint8 * RAM), ("ram_cur", C.c_uint8 * RAM),
                ("stack", C.c_uint8 * OB_BYTES), ("done", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_adam_update.restype = C.c_double
        _lib.orc_sgd_update.restype = C.c_double
        _lib.orc_bc_distance.restype = C.c_double
        _lib.orc_novelty.restype = C.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def layout(kind, nact=18):
    L = Layout()
    lib().orc_layout_make(kind, nact, C.byref(L))
    return L


def num_params(kind, nact=18):
    return lib().orc_num_params(kind, nact)


def perturb(theta, noise, idx, sigma, sign):
    theta = _f32(theta)
    out = np.empty_like(theta)
    lib().orc_perturb(_p(theta, C.c_float), _p(noise, C.c_float), C.c_int64(int(idx)), C.c_float(sigma),
                      int(sign), theta.size, _p(out, C.c_float))
    return out


def es_ref_pass(L, theta, ref):
    theta = _f32(theta)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    bn = np.empty(BN_FLOATS, np.float32)
    lib().orc_es_ref_pass(C.byref(L), _p(theta, C.c_float), _p(ref, C.c_uint8), ref.shape[0], _p(bn, C.c_float))
    return bn


def es_ref_pass_moments(L, theta, ref):
    """(bn scale/shift, batch mean/variance) of the reference pass"""
    theta = _f32(theta)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    bn = np.empty(BN_FLOATS, np.float32); mom = np.empty(BN_FLOATS, np.float32)
    lib().orc_es_ref_pass_moments(C.byref(L), _p(theta, C.c_float), _p(ref, C.c_uint8), ref.shape[0], _p(bn, C.c_float), _p(mom, C.c_float))
    return bn, mom


def act(L, theta, bn, ob):
    theta = _f32(theta)
    ob = np.ascontiguousarray(ob, dtype=np.uint8)
    logits = np.empty(L.nact, np.float32)
    a = lib().orc_act(C.byref(L), _p(theta, C.c_float), _p(bn, C.c_float) if bn is not None else None,
                      _p(ob, C.c_uint8), _p(logits, C.c_float))
    return a, logits


def forward_debug(L, theta, bn, ob):
    theta = _f32(theta)
    ob = np.ascontiguousarray(ob, dtype=np.uint8)
    y1 = np.empty(7056, np.float32); y2 = np.empty(3872, np.float32); y3 = np.empty(256, np.float32)
    lg = np.empty(L.nact, np.float32)
    lib().orc_forward_debug(C.byref(L), _p(theta, C.c_float), _p(bn, C.c_float) if bn is not None else None,
                            _p(ob, C.c_uint8), _p(y1, C.c_float), _p(y2, C.c_float), _p(y3, C.c_float),
                            _p(lg, C.c_float))
    return y1, y2, y3, lg


def forward_large_debug(L, theta, ob):
    """LargeModel of the GPU tree (models/dqn.py:39-47): raw y1 [21,21,32], y2 / y3 [11,11,64], y4 [512], logits"""
    theta = _f32(theta)
    ob = np.ascontiguousarray(ob, dtype=np.uint8)
    y1 = np.empty(21 * 21 * 32, np.float32); y2 = np.empty(11 * 11 * 64, np.float32); y3 = np.empty(11 * 11 * 64, np.float32)
    y4 = np.empty(512, np.float32); lg = np.empty(L.nact, np.float32)
    lib().orc_forward_large_debug(C.byref(L), _p(theta, C.c_float), _p(ob, C.c_uint8), _p(y1, C.c_float), _p(y2, C.c_float),
                                  _p(y3, C.c_float), _p(y4, C.c_float), _p(lg, C.c_float))
    return y1, y2, y3, y4, lg


# ---- SynthAtari raw env -----------------------------------------------------
def raw_reset(seed):
    ram = np.zeros(RAM, np.uint8)
    lib().orc_raw_reset(_p(ram, C.c_uint8), C.c_uint32(seed))
    return ram


def raw_frame(ram, action):
    return lib().orc_raw_frame(_p(ram, C.c_uint8), int(action))


def raw_render(ram):
    scr = np.empty((210, 160), np.uint8)
    lib().orc_raw_render(_p(ram, C.c_uint8), _p(scr, C.c_uint8))
    return scr


def palette():
    pal = np.empty((16, 3), np.uint8)
    lib().orc_palette(_p(pal, C.c_uint8))
    return pal


def warp_rgb(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.empty((84, 84), np.uint8)
    lib().orc_warp_rgb(_p(rgb, C.c_uint8), _p(out, C.c_uint8))
    return out


def resize_tables():
    kh = np.empty((84, 5), np.float64); bh = np.empty((84, 2), np.int32)
    kv = np.empty((84, 7), np.float64); bv = np.empty((84, 2), np.int32)
    lib().orc_resize_tables(_p(kh, C.c_double), _p(bh, C.c_int), _p(kv, C.c_double), _p(bv, C.c_int))
    return kh, bh, kv, bv


class WrappedEnv:
    """wrap_deepmind(SynthAtari) -- atari_wrappers.py:204-222."""

    def __init__(self):
        self.e = WEnv()

    def reset(self, seed):
        lib().orc_wenv_reset(C.byref(self.e), C.c_uint32(int(seed)))
        return self.ob()

    def step(self, action):
        r = C.c_float(); d = C.c_int()
        lib().orc_wenv_step(C.byref(self.e), int(action), C.byref(r), C.byref(d))
        return self.ob(), float(r.value), bool(d.value)

    def ob(self):
        return np.frombuffer(self.e.stack, dtype=np.uint8).reshape(OB_SHAPE).copy()

    def ram(self):
        return np.frombuffer(self.e.ram_cur, dtype=np.uint8).copy()


def get_ref_batch(seed=0, batch_size=128, nact=18, env_seed=0):
    """es.py:105-113 with action_space.sample() drawn from RandomState(seed)."""
    rs = np.random.RandomState(seed)
    env = WrappedEnv()
    env.reset(env_seed)
    out = []
    ep = 0
    while len(out) < batch_size:
        ob, rew, done = env.step(rs.randint(nact))
        out.append(ob)
        if done:
            ep += 1
            env.reset(env_seed + ep)
    return np.stack(out)


def rollout(L, theta, ref, env_seed, tslimit, want_bc=False, want_actions=False):
    theta = _f32(theta)
    ret = C.c_float(); sr = C.c_float(); ln = C.c_int32()
    bc = None
    if want_bc:
        bc = np.zeros((tslimit, RAM) if L.kind == KIND_ES else (RAM,), np.uint8)
    acts = np.zeros(tslimit, np.int32) if want_actions else None
    nref = 0 if ref is None else ref.shape[0]
    lib().orc_rollout(C.byref(L), _p(theta, C.c_float), _p(ref, C.c_uint8) if ref is not None else None, nref,
                      C.c_uint32(int(env_seed)), int(tslimit), C.byref(ret), C.byref(sr), C.byref(ln),
                      _p(bc, C.c_uint8) if bc is not None else None,
                      _p(acts, C.c_int32) if acts is not None else None)
    n = ln.value
    res = [ret.value, sr.value, n]
    if want_bc:
        res.append(bc[:n] if L.kind == KIND_ES else bc)
    if want_actions:
        res.append(acts[:n])
    return tuple(res)


def es_eval(L, theta, noise, idx, sigma, tslimit, ref, env_seed):
    theta = _f32(theta)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    env_seed = np.ascontiguousarray(env_seed, dtype=np.uint32)
    n = idx.size
    rets = np.zeros((n, 2), np.float32); sg = np.zeros((n, 2), np.float32); ln = np.zeros((n, 2), np.int32)
    lib().orc_es_eval(C.byref(L), _p(theta, C.c_float), _p(noise, C.c_float), _p(idx, C.c_int64), n,
                      C.c_float(sigma), int(tslimit), _p(ref, C.c_uint8), ref.shape[0], _p(env_seed, C.c_uint32),
                      _p(rets, C.c_float), _p(sg, C.c_float), _p(ln, C.c_int32))
    return rets, sg, ln


# ---- reduce ------------------------------------------------------------------
def centered_ranks(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_centered_ranks(_p(x, C.c_float), x.size, _p(y, C.c_float))
    return y


def weighted_sum(noise, idx, w, P, denom):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    w = _f32(w)
    g = np.empty(P, np.float32)
    lib().orc_weighted_sum(_p(noise, C.c_float), _p(idx, C.c_int64), _p(w, C.c_float), idx.size, P,
                           C.c_float(denom), _p(g, C.c_float))
    return g


def es_gradient(noise, idx, returns_n2, P):
    """es.py:281-296 for return_proc_mode == 'centered_rank'."""
    proc = centered_ranks(np.asarray(returns_n2, np.float32).reshape(-1)).reshape(-1, 2)
    w = proc[:, 0] - proc[:, 1]
    return weighted_sum(noise, idx, w, P, float(proc.size))


class Adam:
    def __init__(self, theta, stepsize, beta1=0.9, beta2=0.999, epsilon=1e-08):
        self.theta = _f32(theta).copy()
        self.m = np.zeros_like(self.theta); self.v = np.zeros_like(self.theta)
        self.t = 0
        self.args = (stepsize, beta1, beta2, epsilon)

    def update(self, g, l2coeff):
        """theta <- theta + step(-g + l2coeff*theta); returns (ratio, theta)  es.py:298"""
        self.t += 1
        g = _f32(g)
        ratio = lib().orc_adam_update(_p(self.theta, C.c_float), _p(self.m, C.c_float), _p(self.v, C.c_float),
                                      _p(g, C.c_float), g.size, C.c_float(l2coeff), self.t,
                                      *[C.c_double(a) for a in self.args])
        return ratio, self.theta


class SGD:
    def __init__(self, theta, stepsize, momentum=0.9):
        self.theta = _f32(theta).copy()
        self.v = np.zeros_like(self.theta)
        self.args = (stepsize, momentum)

    def update(self, g, l2coeff):
        g = _f32(g)
        ratio = lib().orc_sgd_update(_p(self.theta, C.c_float), _p(self.v, C.c_float), _p(g, C.c_float), g.size,
                                     C.c_float(l2coeff), *[C.c_double(a) for a in self.args])
        return ratio, self.theta


# ---- GA ------------------------------------------------------------------------
def ga_normc(L, theta):
    th = _f32(theta).copy()
    lib().orc_ga_normc(C.byref(L), _p(th, C.c_float))
    return th


def ga_rebuild(L, noise, seeds, sigma):
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    th = np.empty(L.P, np.float32)
    lib().orc_ga_rebuild(C.byref(L), _p(noise, C.c_float), _p(seeds, C.c_int64), seeds.size, C.c_float(sigma),
                         _p(th, C.c_float))
    return th


def ga_select(returns, T):
    returns = _f32(returns)
    out = np.empty(T, np.int32)
    lib().orc_ga_select(_p(returns, C.c_float), returns.size, T, _p(out, C.c_int32))
    return out


# ---- novelty -------------------------------------------------------------------
def bc_distance(x, y):
    x = np.ascontiguousarray(x, np.uint8).reshape(-1, x.shape[-1]); y = np.ascontiguousarray(y, np.uint8).reshape(-1, y.shape[-1])
    return lib().orc_bc_distance(_p(x, C.c_uint8), x.shape[0], _p(y, C.c_uint8), y.shape[0], x.shape[1])


def novelty(archive, bc, k):
    bc = np.ascontiguousarray(bc, np.uint8).reshape(-1, bc.shape[-1])
    arch = [np.ascontiguousarray(a, np.uint8).reshape(-1, bc.shape[1]) for a in archive]
    n = len(arch)
    ptrs = (C.POINTER(C.c_uint8) * n)(*[_p(a, C.c_uint8) for a in arch])
    lens = (C.c_int * n)(*[a.shape[0] for a in arch])
    return lib().orc_novelty(ptrs, lens, n, _p(bc, C.c_uint8), bc.shape[0], bc.shape[1], int(k))


# ---- initial parameters (Xavier-uniform weights drawn with RandomState(seed)) -----
def es_init_theta(L, seed=0):
    """tf.contrib.layers defaults: xavier_initializer (uniform, limit sqrt(6/(fan_in+fan_out))),
    zero biases, BN beta=0 gamma=1  [external: TF defaults; TF's own init is unseeded]"""
    rs = np.random.RandomState(seed)
    th = np.zeros(L.P, np.float32)

    def xav(off, shape):
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        n = int(np.prod(shape))
        th[off:off + n] = rs.uniform(-lim, lim, n).astype(np.float32)

    xav(L.c1w, (8, 8, 4, 16)); xav(L.c2w, (4, 4, 16, 32)); xav(L.fcw, (3872, 256)); xav(L.ow, (256, L.nact))
    th[L.bn1g:L.bn1g + 16] = 1; th[L.bn2g:L.bn2g + 32] = 1; th[L.bn3g:L.bn3g + 256] = 1
    return th


def ga_gpu_rebuild(noise, seeds, scale_by):
    """gpu_implementation/neuroevolution/models/base.py:118-149 (compute_weights_from_seeds / compute_mutation), float32:
    theta = noise.get(idx0, P).copy() * scale_by; for (idx, power) in seeds[1:]: theta = theta + power * noise.get(idx, P).
    (`power * noise` with a Python float scalar stays float32 under numpy's value-based casting, like es.py:413.)"""
    scale_by = np.asarray(scale_by, np.float32)
    P = scale_by.size
    idx0 = seeds[0][0] if isinstance(seeds[0], (tuple, list)) else seeds[0]
    theta = noise[idx0:idx0 + P].copy() * scale_by
    for idx, power in seeds[1:]:
        theta = theta + np.float32(power) * noise[idx:idx + P]
    return theta.astype(np.float32)
