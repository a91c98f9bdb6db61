"""ctypes shim over libdne_hip.so (include/dne_hip.h) -- the only way Python reaches the HIP kernels.

There is no CPU fallback: if the shared library is missing, or no MI355X is visible, constructing an
Engine raises.  numpy arrays cross the boundary as plain pointers + sizes.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_PATH = os.environ.get("DNE_LIB_PATH") or os.path.join(_CSRC, "libdne_hip.so")   # DNE_LIB_PATH: another build of the same ABI (same-box A/B of whole builds)

KIND_ES, KIND_GA, KIND_GA_LARGE = 0, 1, 2   # DNE_KIND_* (include/dne_hip.h); 2 = the GPU tree's LargeModel (models/dqn.py:39-47)
PROC_MODES = {"centered_rank": 0, "sign": 1, "centered_sign_rank": 2}
OPT_KINDS = {"adam": 0, "sgd": 1}
OB_SHAPE = (84, 84, 4)
OB_BYTES = 84 * 84 * 4
RAM_BYTES = 128
BN_FLOATS = 608
# gym registers *NoFrameskip-v4 with max_episode_steps = 400000, counted by its TimeLimit wrapper in steps of the RAW environment
# (frames).  policies.py:383-385 reads that number through env.spec and applies it as a bound on AGENT steps (one per 4 frames),
# while the real TimeLimit sits inside the wrappers and would end the episode after 100000 agent steps.  The mirror keeps the
# reference's arithmetic as written (min(task limit, 400000) agent steps): 4x looser than gym's own limit, never reached by the
# BASELINE configurations (tslimit 5000) -- DESIGN.md section 5.
ENV_MAX_EPISODE_STEPS = 400000


class DneError(RuntimeError):
    pass


# one (noise_idx, returns, lengths, sign-returns) record per antithetic pair -- what travels between GPUs;
# the layout of struct PairRecord in csrc/reduce.h
RECORD = np.dtype([('noise_idx', '<i8'), ('ret', '<f4', (2,)), ('len', '<i4', (2,)), ('aux', '<f4', (2,))])
assert RECORD.itemsize == 32


def comm_unique_id():
    """ncclGetUniqueId through the C ABI (rank 0 calls it and hands the 128 bytes to every rank)"""
    buf = (C.c_char * 128)()
    if load().dne_comm_unique_id(buf) != 0:
        raise DneError(load().dne_last_error(None).decode())
    return bytes(buf.raw)


def device_count():
    """HIP devices visible to this process (0 without a GPU)"""
    n = C.c_int(0)
    if load().dne_device_count(C.byref(n)) != 0:
        raise DneError(load().dne_last_error(None).decode())
    return n.value


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("device_id", "policy_kind", "n_actions", "max_members", "ref_count",
                                         "ref_chunk", "record_bc", "bc_max_steps", "profile_events", "bc_final_only")] + \
               [("reserved", C.c_int32 * 6)]


class Profile(C.Structure):
    _fields_ = [("eval_ms", C.c_double), ("fc_ms", C.c_double), ("fc_launches", C.c_int64),
                ("fc_group_steps", C.c_int64), ("env_steps", C.c_int64), ("conv_ms", C.c_double),
                ("env_ms", C.c_double), ("ref_ms", C.c_double), ("reduce_ms", C.c_double),
                ("materialize_ms", C.c_double), ("fc_full_ms", C.c_double), ("fc_full_launches", C.c_double),
                ("fc_full_units", C.c_double), ("fc_full_kind", C.c_double), ("fc_full_union_ms", C.c_double), ("reserved", C.c_double * 1)]


def build(force=False):
    """Compile libdne_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in ("engine.hip", "forward.h", "forward_variants.h", "forward_large.h", "reduce.h", "env_synth.h")]
    srcs.append(os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "include", "dne_hip.h"))
    if os.environ.get("DNE_LIB_PATH"):
        # another build of the same ABI was asked for by name: `make` only knows the in-tree library, so running it here would
        # rebuild THAT one in the middle of an A/B and still not produce the file named -- check the file and leave it alone
        if not os.path.exists(LIB_PATH):
            raise DneError("DNE_LIB_PATH=%s does not exist (it names a prebuilt library; nothing is built for it)" % LIB_PATH)
        return LIB_PATH
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _CSRC, "-s"])
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DneError("libdne_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "-- the HIP engine has no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        lib.dne_last_error.restype = C.c_char_p
        lib.dne_last_error.argtypes = [C.c_void_p]
        lib.dne_destroy.restype = None
        _lib = lib
    return _lib


def num_params(kind, nact=18):
    return load().dne_num_params(int(kind), int(nact))


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Engine:
    """One engine = one GPU.  Thin, typed wrapper; every method maps 1:1 onto a dne_* entry point."""

    def __init__(self, kind, n_actions=18, max_members=256, ref_count=128, device_id=0, ref_chunk=0,
                 record_bc=False, bc_max_steps=0, profile_events=False, bc_final_only=False):
        self.lib = load()
        self.kind, self.n_actions, self.max_members = int(kind), int(n_actions), int(max_members)
        self.ref_count = int(ref_count) if kind == KIND_ES else 0
        self.bc_max_steps = int(bc_max_steps)
        cfg = Config(device_id=device_id, policy_kind=self.kind, n_actions=self.n_actions, max_members=self.max_members,
                     ref_count=self.ref_count, ref_chunk=ref_chunk, record_bc=int(bool(record_bc)),
                     bc_max_steps=self.bc_max_steps, profile_events=int(bool(profile_events)),
                     bc_final_only=int(bool(bc_final_only)))
        self.bc_final_only = bool(bc_final_only)
        self.h = C.c_void_p()
        rc = self.lib.dne_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            raise DneError(self.lib.dne_last_error(None).decode())
        self.P = self.lib.dne_num_params(self.kind, self.n_actions)
        self.noise_count = 0
        self.comm_size = 1

    def close(self):
        if getattr(self, "h", None):
            self.lib.dne_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise DneError(self.lib.dne_last_error(self.h).decode())

    # ---- noise / parameters
    def noise_upload(self, noise):
        noise = _arr(noise, np.float32)
        self._ck(self.lib.dne_noise_upload(self.h, _ptr(noise, C.c_float), C.c_size_t(noise.size)))
        self.noise_count = noise.size

    def noise_alloc(self, count):
        self._ck(self.lib.dne_noise_alloc(self.h, C.c_size_t(int(count))))
        self.noise_count = int(count)

    def noise_write(self, offset, chunk):
        chunk = _arr(chunk, np.float32)
        self._ck(self.lib.dne_noise_write(self.h, C.c_size_t(int(offset)), _ptr(chunk, C.c_float), C.c_size_t(chunk.size)))

    def check_redzones(self):
        """0 = no kernel wrote outside its device buffer; raises with the buffer's name otherwise"""
        rc = self.lib.dne_check_redzones(self.h)
        if rc != 0:
            raise DneError(self.lib.dne_last_error(self.h).decode())
        return 0

    def noise_get(self, idx, dim):
        out = np.empty(dim, np.float32)
        self._ck(self.lib.dne_noise_get(self.h, C.c_int64(int(idx)), int(dim), _ptr(out, C.c_float)))
        return out

    def set_theta(self, theta, slot=0):
        theta = _arr(theta, np.float32)
        self._ck(self.lib.dne_set_theta(self.h, int(slot), _ptr(theta, C.c_float), C.c_size_t(theta.size)))

    def get_theta(self, slot=0):
        out = np.empty(self.P, np.float32)
        self._ck(self.lib.dne_get_theta(self.h, int(slot), _ptr(out, C.c_float), C.c_size_t(out.size)))
        return out

    def set_ref_batch(self, ref):
        ref = _arr(ref, np.uint8)
        assert ref.shape[1:] == OB_SHAPE
        self._ck(self.lib.dne_set_ref_batch(self.h, _ptr(ref, C.c_uint8), int(ref.shape[0])))

    def materialize(self, idx, sigma, copy_out=True):
        idx = _arr(idx, np.int64)
        out = np.empty((idx.size, 2, self.P), np.float32) if copy_out else None
        self._ck(self.lib.dne_materialize(self.h, _ptr(idx, C.c_int64), int(idx.size), C.c_float(sigma), _ptr(out, C.c_float)))
        return out

    # ---- env
    def env_reset(self, seeds):
        seeds = _arr(seeds, np.uint32)
        self._ck(self.lib.dne_env_reset(self.h, int(seeds.size), _ptr(seeds, C.c_uint32)))

    def env_step(self, actions):
        actions = _arr(actions, np.int32)
        n = actions.size
        rew = np.empty(n, np.float32); done = np.empty(n, np.int32)
        self._ck(self.lib.dne_env_step(self.h, n, _ptr(actions, C.c_int32), _ptr(rew, C.c_float), _ptr(done, C.c_int32)))
        return rew, done.astype(bool)

    def env_observation(self, n):
        out = np.empty((n,) + OB_SHAPE, np.uint8)
        self._ck(self.lib.dne_env_observation(self.h, int(n), _ptr(out, C.c_uint8)))
        return out

    def env_ram(self, n):
        out = np.empty((n, RAM_BYTES), np.uint8)
        self._ck(self.lib.dne_env_ram(self.h, int(n), _ptr(out, C.c_uint8)))
        return out

    def env_set_observation(self, obs):
        obs = _arr(obs, np.uint8)
        self._ck(self.lib.dne_env_set_observation(self.h, int(obs.shape[0]), _ptr(obs, C.c_uint8)))

    def env_set_ram(self, ram_prev, ram_cur):
        """inject emulator states (RAM before / after the last raw frame); observations are re-rendered as after a reset"""
        ram_prev = _arr(ram_prev, np.uint8).reshape(-1, RAM_BYTES); ram_cur = _arr(ram_cur, np.uint8).reshape(-1, RAM_BYTES)
        assert ram_prev.shape == ram_cur.shape
        self._ck(self.lib.dne_env_set_ram(self.h, int(ram_prev.shape[0]), _ptr(ram_prev, C.c_uint8), _ptr(ram_cur, C.c_uint8)))

    # ---- forward
    def set_members(self, slot, off, scale):
        slot = _arr(slot, np.int32); off = _arr(off, np.int64); scale = _arr(scale, np.float32)
        assert slot.size == off.size == scale.size
        self._ck(self.lib.dne_set_members(self.h, int(slot.size), _ptr(slot, C.c_int32), _ptr(off, C.c_int64), _ptr(scale, C.c_float)))

    def ref_pass(self, n):
        self._ck(self.lib.dne_ref_pass(self.h, int(n)))

    def get_bn(self, n):
        out = np.empty((n, BN_FLOATS), np.float32)
        self._ck(self.lib.dne_get_bn(self.h, int(n), _ptr(out, C.c_float)))
        return out

    def get_bn_moments(self, n):
        out = np.empty((n, BN_FLOATS), np.float32)
        self._ck(self.lib.dne_get_bn_moments(self.h, int(n), _ptr(out, C.c_float)))
        return out

    def act(self, n):
        actions = np.empty(n, np.int32); logits = np.empty((n, self.n_actions), np.float32)
        self._ck(self.lib.dne_act(self.h, int(n), _ptr(actions, C.c_int32), _ptr(logits, C.c_float)))
        return actions, logits

    def debug_activations_large(self, member):
        """LargeModel: raw conv1 [441*32], conv2 / conv3 [121*64] and fc [512] outputs of one member after act()"""
        y1 = np.empty(14112, np.float32); y2 = np.empty(7744, np.float32); y3 = np.empty(7744, np.float32); y4 = np.empty(512, np.float32)
        self._ck(self.lib.dne_debug_activations_large(self.h, int(member), _ptr(y1, C.c_float), _ptr(y2, C.c_float), _ptr(y3, C.c_float), _ptr(y4, C.c_float)))
        return y1, y2, y3, y4

    def debug_activations(self, member):
        y1 = np.empty(7056, np.float32); y2 = np.empty(3872, np.float32); y3 = np.empty(256, np.float32)
        self._ck(self.lib.dne_debug_activations(self.h, int(member), _ptr(y1, C.c_float), _ptr(y2, C.c_float), _ptr(y3, C.c_float)))
        return y1, y2, y3

    # ---- batch evaluation
    def _bc_buf(self, n, want):
        if not want:
            return None
        if self.kind == KIND_ES and not self.bc_final_only:
            return np.zeros((n, self.bc_max_steps, RAM_BYTES), np.uint8)
        return np.zeros((n, RAM_BYTES), np.uint8)

    def es_eval(self, noise_idx, sigma, tslimit, env_seed, want_bc=False):
        idx = _arr(noise_idx, np.int64); seeds = _arr(env_seed, np.uint32)
        n = idx.size
        assert seeds.size == 2 * n
        ret = np.empty((n, 2), np.float32); sg = np.empty((n, 2), np.float32); ln = np.empty((n, 2), np.int32)
        bc = self._bc_buf(2 * n, want_bc)
        self._ck(self.lib.dne_es_eval(self.h, _ptr(idx, C.c_int64), n, C.c_float(sigma), int(tslimit), _ptr(seeds, C.c_uint32),
                                      _ptr(ret, C.c_float), _ptr(sg, C.c_float), _ptr(ln, C.c_int32), _ptr(bc, C.c_uint8)))
        return (ret, sg, ln, bc) if want_bc else (ret, sg, ln)

    def eval_members(self, n, tslimit, env_seed, want_bc=False):
        seeds = _arr(env_seed, np.uint32)
        assert seeds.size == n
        ret = np.empty(n, np.float32); sg = np.empty(n, np.float32); ln = np.empty(n, np.int32)
        bc = self._bc_buf(n, want_bc)
        self._ck(self.lib.dne_eval_members(self.h, int(n), int(tslimit), _ptr(seeds, C.c_uint32), _ptr(ret, C.c_float),
                                           _ptr(sg, C.c_float), _ptr(ln, C.c_int32), _ptr(bc, C.c_uint8)))
        return (ret, sg, ln, bc) if want_bc else (ret, sg, ln)

    def ga_eval(self, chains, sigma, tslimit, env_seed, want_bc=False):
        n = len(chains)
        co = np.zeros(n + 1, np.int32)
        co[1:] = np.cumsum([len(c) for c in chains])
        flat = _arr(np.concatenate([np.asarray(c, np.int64) for c in chains]), np.int64)
        seeds = _arr(env_seed, np.uint32)
        assert seeds.size == n
        ret = np.empty(n, np.float32); sg = np.empty(n, np.float32); ln = np.empty(n, np.int32)
        bc = self._bc_buf(n, want_bc)
        self._ck(self.lib.dne_ga_eval(self.h, _ptr(co, C.c_int32), _ptr(flat, C.c_int64), n, C.c_float(sigma), int(tslimit),
                                      _ptr(seeds, C.c_uint32), _ptr(ret, C.c_float), _ptr(sg, C.c_float), _ptr(ln, C.c_int32),
                                      _ptr(bc, C.c_uint8)))
        return (ret, sg, ln, bc) if want_bc else (ret, sg, ln)

    # ---- gpu-tree genomes: ((idx0,), (idx1, power1), ...)
    def ga_set_init_scale(self, scale_by):
        sb = _arr(scale_by, np.float32)
        self._ck(self.lib.dne_ga_set_init_scale(self.h, _ptr(sb, C.c_float), C.c_size_t(sb.size)))

    @staticmethod
    def _split_powers(chain):
        idx = np.array([c[0] if isinstance(c, (tuple, list)) else c for c in chain], np.int64)
        pw = np.array([c[1] if isinstance(c, (tuple, list)) and len(c) > 1 else 0.0 for c in chain], np.float32)
        return idx, pw

    def ga_rebuild_powers(self, slot, seeds, copy_out=True):
        idx, pw = self._split_powers(seeds)
        out = np.empty(self.P, np.float32) if copy_out else None
        self._ck(self.lib.dne_ga_rebuild_powers(self.h, int(slot), _ptr(idx, C.c_int64), _ptr(pw, C.c_float), int(idx.size), _ptr(out, C.c_float)))
        return out

    def ga_eval_powers(self, genomes, tslimit, env_seed, want_bc=False):
        n = len(genomes)
        co = np.zeros(n + 1, np.int32)
        co[1:] = np.cumsum([len(g) for g in genomes])
        parts = [self._split_powers(g) for g in genomes]
        flat = _arr(np.concatenate([p[0] for p in parts]), np.int64)
        pw = _arr(np.concatenate([p[1] for p in parts]), np.float32)
        seeds = _arr(env_seed, np.uint32)
        assert seeds.size == n
        ret = np.empty(n, np.float32); sg = np.empty(n, np.float32); ln = np.empty(n, np.int32)
        bc = self._bc_buf(n, want_bc)
        self._ck(self.lib.dne_ga_eval_powers(self.h, _ptr(co, C.c_int32), _ptr(flat, C.c_int64), _ptr(pw, C.c_float), n, int(tslimit),
                                             _ptr(seeds, C.c_uint32), _ptr(ret, C.c_float), _ptr(sg, C.c_float), _ptr(ln, C.c_int32),
                                             _ptr(bc, C.c_uint8)))
        return (ret, sg, ln, bc) if want_bc else (ret, sg, ln)

    def ga_rebuild(self, slot, seeds, sigma, copy_out=True):
        seeds = _arr(seeds, np.int64)
        out = np.empty(self.P, np.float32) if copy_out else None
        self._ck(self.lib.dne_ga_rebuild(self.h, int(slot), _ptr(seeds, C.c_int64), int(seeds.size), C.c_float(sigma), _ptr(out, C.c_float)))
        return out

    # ---- reduce
    def centered_ranks(self, x):
        x = _arr(x, np.float32)
        out = np.empty(x.size, np.float32)
        self._ck(self.lib.dne_centered_ranks(self.h, _ptr(x.reshape(-1), C.c_float), int(x.size), _ptr(out, C.c_float)))
        return out.reshape(x.shape)

    def weighted_sum(self, idx, w, denom, copy_out=True):
        idx = _arr(idx, np.int64); w = _arr(w, np.float32)
        g = np.empty(self.P, np.float32) if copy_out else None
        self._ck(self.lib.dne_weighted_sum(self.h, _ptr(idx, C.c_int64), _ptr(w, C.c_float), int(idx.size), C.c_float(denom), _ptr(g, C.c_float)))
        return g

    def optimizer_reset(self):
        self._ck(self.lib.dne_optimizer_reset(self.h))

    def optimizer_get_state(self):
        m = np.empty(self.P, np.float32); v = np.empty(self.P, np.float32); t = C.c_int32()
        self._ck(self.lib.dne_optimizer_get_state(self.h, _ptr(m, C.c_float), _ptr(v, C.c_float), C.byref(t)))
        return m, v, t.value

    def optimizer_set_state(self, m, v, t):
        m = _arr(m, np.float32); v = _arr(v, np.float32)
        self._ck(self.lib.dne_optimizer_set_state(self.h, _ptr(m, C.c_float), _ptr(v, C.c_float), int(t)))

    def optimizer_step(self, kind, l2coeff, stepsize, beta1_or_momentum=0.9, beta2=0.999, epsilon=1e-8):
        ratio = C.c_double()
        self._ck(self.lib.dne_optimizer_step(self.h, OPT_KINDS[kind], C.c_float(l2coeff), C.c_double(stepsize),
                                             C.c_double(beta1_or_momentum), C.c_double(beta2), C.c_double(epsilon), C.byref(ratio)))
        return ratio.value

    def es_update(self, idx, returns_n2, signreturns_n2, proc_mode, opt_kind, l2coeff, stepsize,
                  beta1_or_momentum=0.9, beta2=0.999, epsilon=1e-8):
        idx = _arr(idx, np.int64); r = _arr(returns_n2, np.float32)
        s = _arr(signreturns_n2, np.float32) if signreturns_n2 is not None else None
        ratio = C.c_double()
        self._ck(self.lib.dne_es_update(self.h, _ptr(idx, C.c_int64), _ptr(r, C.c_float), _ptr(s, C.c_float), int(idx.size),
                                        PROC_MODES[proc_mode], OPT_KINDS[opt_kind], C.c_float(l2coeff), C.c_double(stepsize),
                                        C.c_double(beta1_or_momentum), C.c_double(beta2), C.c_double(epsilon), C.byref(ratio)))
        return ratio.value

    # ---- exchange between GPUs (RCCL behind the C ABI; the host transports use pack / set)
    def comm_init(self, rank, nranks, unique_id):
        assert len(unique_id) == 128
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        rc = self.lib.dne_comm_init(self.h, int(rank), int(nranks), buf)
        if rc == -2:   # DNE_COMM_DROPPED: the late path must not touch the handle's error text (another thread owns the handle by now)
            raise DneError("dne_comm_init: communicator dropped, dne_comm_abort was called while ncclCommInitRank was in flight")
        self._ck(rc)
        self.comm_size = int(nranks)

    def comm_share(self, owner):
        """take part in the RCCL communicator another engine of this process (same device) built with comm_init"""
        self._ck(self.lib.dne_comm_share(self.h, owner.h))
        self.comm_size = owner.comm_size

    def comm_abort(self):
        self._ck(self.lib.dne_comm_abort(self.h))
        self.comm_size = 1

    def comm_info(self):
        """(rank, nranks, is_rccl) as the communicator itself reports them (ncclCommUserRank / ncclCommCount)"""
        r, n, k = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.lib.dne_comm_info(self.h, C.byref(r), C.byref(n), C.byref(k)))
        return r.value, n.value, bool(k.value)

    def comm_allreduce(self, values, op="sum"):
        v = np.array(values, np.float64).reshape(-1)
        self._ck(self.lib.dne_comm_allreduce(self.h, _ptr(v, C.c_double), int(v.size), {"sum": 0, "max": 1}[op]))
        return v

    def barrier(self):
        self._ck(self.lib.dne_comm_allreduce(self.h, None, 0, 0))

    def comm_allgather(self, arr):
        """every rank's `arr` (same shape and dtype everywhere) stacked in rank order"""
        arr = np.ascontiguousarray(arr)
        world = self.comm_size
        out = np.empty((world,) + arr.shape, arr.dtype)
        self._ck(self.lib.dne_comm_allgather(self.h, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes),
                                             out.ctypes.data_as(C.c_void_p)))
        return out

    def allgather_results(self, n_local, n_global):
        rec = np.zeros(int(n_global), RECORD)
        self._ck(self.lib.dne_allgather_results(self.h, int(n_local), int(n_global), rec.ctypes.data_as(C.c_void_p)))
        return rec

    def debug_unshard(self, gathered, n_global, world):
        g = np.ascontiguousarray(gathered, RECORD)
        out = np.zeros(int(n_global), RECORD)
        self._ck(self.lib.dne_debug_unshard(self.h, g.ctypes.data_as(C.c_void_p), int(n_global), int(world), out.ctypes.data_as(C.c_void_p)))
        return out

    def records_pack(self, n_local):
        rec = np.zeros(int(n_local), RECORD)
        self._ck(self.lib.dne_records_pack(self.h, int(n_local), rec.ctypes.data_as(C.c_void_p)))
        return rec

    def records_set(self, rec):
        rec = np.ascontiguousarray(rec, RECORD)
        self._ck(self.lib.dne_records_set(self.h, rec.ctypes.data_as(C.c_void_p), int(rec.size)))

    def es_update_gathered(self, proc_mode, opt_kind, l2coeff, stepsize, beta1_or_momentum=0.9, beta2=0.999, epsilon=1e-8):
        ratio = C.c_double()
        self._ck(self.lib.dne_es_update_gathered(self.h, PROC_MODES[proc_mode], OPT_KINDS[opt_kind], C.c_float(l2coeff),
                                                 C.c_double(stepsize), C.c_double(beta1_or_momentum), C.c_double(beta2),
                                                 C.c_double(epsilon), C.byref(ratio)))
        return ratio.value

    def ga_select(self, returns, t):
        r = _arr(returns, np.float32)
        out = np.empty(t, np.int32)
        self._ck(self.lib.dne_ga_select(self.h, _ptr(r, C.c_float), int(r.size), int(t), _ptr(out, C.c_int32)))
        return out

    def _sync_archive(self, archive):
        """Bring the device-resident archive up to `archive`.  The master's archive only grows (dist.py:93-98) and the
        transport hands back the same entry objects on every call, so entries are recognised by identity and only new ones
        are uploaded; any other list (fresh arrays, a shorter archive) is uploaded from scratch."""
        have = getattr(self, "_arch_objs", [])
        if len(have) > len(archive) or any(a is not b for a, b in zip(have, archive)):
            self._ck(self.lib.dne_archive_clear(self.h))
            have = []
        for a in archive[len(have):]:
            u = _arr(a, np.uint8)
            u = u.reshape(-1, u.shape[-1])
            self._ck(self.lib.dne_archive_append(self.h, _ptr(u, C.c_uint8), int(u.shape[0]), int(u.shape[1])))
        self._arch_objs = list(archive)      # references keep the identities from being recycled

    def novelty(self, archive, bc, k):
        bc = _arr(bc, np.uint8).reshape(-1, np.asarray(bc).shape[-1])
        self._sync_archive(archive)
        out = C.c_double()
        self._ck(self.lib.dne_novelty(self.h, None, None, 0, _ptr(bc, C.c_uint8), int(bc.shape[0]), int(bc.shape[1]), int(k),
                                      C.byref(out)))
        return out.value

    def novelty_batch(self, archive, lengths, k):
        """novelty of every member's RAM trajectory recorded by the last es_eval (kept on the device)"""
        self._sync_archive(archive)
        ln = _arr(np.asarray(lengths).reshape(-1), np.int32)
        out = np.empty(ln.size, np.float64)
        self._ck(self.lib.dne_novelty_batch(self.h, None, None, 0, int(ln.size), _ptr(ln, C.c_int32), int(k), _ptr(out, C.c_double)))
        return out

    def profile(self):
        p = Profile()
        self._ck(self.lib.dne_get_profile(self.h, C.byref(p)))
        return {f[0]: getattr(p, f[0]) for f in Profile._fields_ if f[0] != "reserved"}
