"""Host-side mirror of es_distributed/policies.py for the Atari policies, backed by the HIP engine.

Keeps the reference's Policy surface (policies.py:15-113, 305-513): num_params, set_trainable_flat /
get_trainable_flat, set_ref_batch, needs_ob_stat / needs_ref_batch, reinitialize, act, rollout, save / Load.
The network itself never exists on the host: parameters are a flat float32 vector in the creation order
of the reference's trainable variables (flat_layout below), the forward runs in libdne_hip.so.
"""
import pickle
from collections import OrderedDict

import numpy as np

from . import _lib


def flat_layout(kind, nact):
    """Offsets of each trainable variable in the flat vector (policies.py:21-24, tf_util.py:224-246).
    ESAtariPolicy: tf.contrib.layers order (weights, biases, BN beta, BN gamma per layer), policies.py:319-330.
    GAAtariPolicy: U.conv / U.dense order (w, b), policies.py:449-459 via tf_util.py:133-162."""
    spec = OrderedDict()
    o = 0

    def add(name, shape):
        nonlocal o
        spec[name] = (o, tuple(shape))
        o += int(np.prod(shape))

    if kind == _lib.KIND_ES:
        add("conv1/weights", (8, 8, 4, 16)); add("conv1/biases", (16,)); add("BatchNorm/beta", (16,)); add("BatchNorm/gamma", (16,))
        add("conv2/weights", (4, 4, 16, 32)); add("conv2/biases", (32,)); add("BatchNorm_1/beta", (32,)); add("BatchNorm_1/gamma", (32,))
        add("fc/weights", (3872, 256)); add("fc/biases", (256,)); add("BatchNorm_2/beta", (256,)); add("BatchNorm_2/gamma", (256,))
        add("out/weights", (256, nact)); add("out/biases", (nact,))
    elif kind == _lib.KIND_GA_LARGE:   # the GPU tree's LargeModel, gpu_implementation/neuroevolution/models/dqn.py:39-47 (creation order, base.py:35-41)
        add("conv1/w", (8, 8, 4, 32)); add("conv1/b", (1, 1, 1, 32))
        add("conv2/w", (4, 4, 32, 64)); add("conv2/b", (1, 1, 1, 64))
        add("conv3/w", (3, 3, 64, 64)); add("conv3/b", (1, 1, 1, 64))
        add("fc/w", (7744, 512)); add("fc/b", (1, 512))
        add("out/w", (512, nact)); add("out/b", (1, nact))
    else:
        add("conv1/w", (8, 8, 4, 16)); add("conv1/b", (1, 1, 1, 16))
        add("conv2/w", (4, 4, 16, 32)); add("conv2/b", (1, 1, 1, 32))
        add("fc/w", (3872, 256)); add("fc/b", (256,))
        add("out/w", (256, nact)); add("out/b", (nact,))
    return spec, o


def xavier_flat(nact, seed=0):
    """Initial ESAtariPolicy parameters: tf.contrib.layers defaults (xavier-uniform weights, zero biases,
    BN beta 0 / gamma 1) drawn from RandomState(seed) -- TF's own initialiser is unseeded."""
    spec, P = flat_layout(_lib.KIND_ES, nact)
    rs = np.random.RandomState(seed)
    th = np.zeros(P, np.float32)
    for name, (off, shape) in spec.items():
        n = int(np.prod(shape))
        if name.endswith("weights"):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            lim = np.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
            th[off:off + n] = rs.uniform(-lim, lim, n).astype(np.float32)
        elif name.endswith("gamma"):
            th[off:off + n] = 1.0
    return th


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class HipAtariEnv:
    """The engine's batched environment seen through the gym-style surface the reference drivers use
    (es.py:131-133: gym.make + wrap_deepmind).  One instance = env slot 0 of an Engine."""
    observation_space = _Space(shape=_lib.OB_SHAPE)

    def __init__(self, engine, seed=0):
        self.engine = engine
        self.action_space = _Space(n=engine.n_actions)
        self._episode = 0
        self._seed = int(seed)
        self.np_random = np.random.RandomState(seed)

    def seed(self, seed):
        self._seed, self._episode = int(seed), 0
        self.np_random = np.random.RandomState(seed)

    def next_episode_seed(self):
        s = (self._seed + self._episode) & 0xFFFFFFFF
        self._episode += 1
        return s

    def reset(self):
        self.engine.env_reset(np.array([self.next_episode_seed()], np.uint32))
        return self._ob()

    def step(self, action):
        rew, done = self.engine.env_step(np.array([action], np.int32))
        return self._ob(), float(rew[0]), bool(done[0]), {}

    def _ob(self):  # ScaledFloatFrame: float32(u8) / 255.0  (atari_wrappers.py:183-186)
        return self.engine.env_observation(1)[0].astype(np.float32) / np.float32(255.0)

    def _get_ram(self):
        return self.engine.env_ram(1)[0]

    @property
    def unwrapped(self):
        return self


class Policy:
    kind = None

    def __init__(self, ob_space, ac_space, engine=None, **kwargs):
        self.args, self.kwargs = (ob_space, ac_space), kwargs
        self.ob_space_shape = tuple(ob_space.shape)
        self.ac_space = ac_space
        self.num_actions = ac_space.n
        self.spec, self.num_params = flat_layout(self.kind, self.num_actions)
        self.engine = engine
        self._flat = np.zeros(self.num_params, np.float32)
        assert self.num_params == _lib.num_params(self.kind, self.num_actions)

    # policies.py:99-103
    def set_trainable_flat(self, x):
        x = np.asarray(x, np.float32)
        assert x.shape == (self.num_params,)
        self._flat = x.copy()
        if self.engine is not None:
            self.engine.set_theta(self._flat, slot=0)

    def get_trainable_flat(self):
        if self.engine is not None:
            self._flat = self.engine.get_theta(0)
        return self._flat.copy()

    @property
    def needs_ob_stat(self):
        return False

    def act(self, ob, random_stream=None):
        """policies.py:374-375 / 469-470: argmax action for a batch of float observations in [0, 1]."""
        obs = np.asarray(ob[0] if isinstance(ob, (list, tuple)) else ob, np.float32)
        u8 = np.rint(obs * 255.0).astype(np.uint8).reshape((-1,) + _lib.OB_SHAPE)
        n = u8.shape[0]
        e = self.engine
        e.env_set_observation(u8)
        e.set_members(np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.float32))
        if self.kind == _lib.KIND_ES:
            e.ref_pass(n)
        return e.act(n)[0].astype(np.int64)

    def rollout(self, env, *, render=False, timestep_limit=None, save_obs=False, random_stream=None,
                worker_stats=None, policy_seed=None):
        """policies.py:378-429 / 473-513: one episode of the current flat parameters.  Returns
        (rews float32[T], T, novelty_vector) like the reference (ES: RAM per step, GA: final RAM)."""
        e = self.engine
        limit = _lib.ENV_MAX_EPISODE_STEPS if timestep_limit is None else min(timestep_limit, _lib.ENV_MAX_EPISODE_STEPS)
        if policy_seed:
            env.seed(policy_seed)
        seed = env.next_episode_seed()
        e.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
        # a single episode needs per-step rewards: drive it step by step through the env ABI
        e.env_reset(np.array([seed], np.uint32))
        if self.kind == _lib.KIND_ES:
            e.ref_pass(1)
        rews, rams, obs = [], [], []
        for _ in range(limit):
            if save_obs:
                obs.append(e.env_observation(1)[0].astype(np.float32) / np.float32(255.0))
            a = e.act(1)[0]
            rew, done = e.env_step(a)
            rews.append(rew[0])
            if self.kind == _lib.KIND_ES:
                rams.append(e.env_ram(1)[0])
            if done[0]:
                break
        nov = np.array(rams) if self.kind == _lib.KIND_ES else e.env_ram(1)[0]
        rews = np.array(rews, dtype=np.float32)
        if save_obs:
            return rews, len(rews), np.array(obs), nov
        return rews, len(rews), nov

    # ---- snapshots, policies.py:49-67
    def variable_arrays(self):
        """{TF variable name: array} for every variable the reference's Policy.save writes (policies.py:52-53:
        `for v in self.all_variables: f[v.name] = v.eval()`), e.g. 'ESAtariPolicy/conv1/weights:0'."""
        flat = self.get_trainable_flat()
        scope = type(self).__name__
        out = OrderedDict()
        for name, (off, shape) in self.spec.items():
            out['%s/%s:0' % (scope, name)] = flat[off:off + int(np.prod(shape))].reshape(shape).copy()
        out.update(self.extra_variable_arrays(scope))
        return out

    def extra_variable_arrays(self, scope):
        return {}

    def save(self, filename):
        """policies.py:49-57.  '.h5': the reference's HDF5 layout -- a float32 dataset per TF variable name, root attributes
        'name' and 'args_and_kwargs' -- written through h5py when it is installed, else through libhdf5 directly
        (h5lite.py; this image has the C library but not h5py).  Any other name: the same arrays, names and attributes in
        a numpy container, which tools/npz_to_h5.py turns into the .h5 (snapshot_extension() picks what the drivers use)."""
        arrays = self.variable_arrays()
        blob = _dumps_spaces(tuple(self.ob_space_shape), self.num_actions, self.kwargs) if filename.endswith('.h5') else \
            pickle.dumps((((tuple(self.ob_space_shape), self.num_actions)), self.kwargs), protocol=-1)
        if filename.endswith('.h5'):
            try:
                import h5py
            except ImportError:
                h5py = None
            if h5py is not None:
                with h5py.File(filename, 'w', libver='latest') as f:
                    for k, v in arrays.items():
                        f[k] = v
                    f.attrs['name'] = type(self).__name__
                    f.attrs['args_and_kwargs'] = np.void(blob)
                return
            from . import h5lite
            if not h5lite.available():
                raise RuntimeError("writing '.h5' snapshots needs h5py or libhdf5 (neither found; DNE_HDF5_LIB names the "
                                   "library): save to '.npz' and run tools/npz_to_h5.py where h5py exists")
            h5lite.write_snapshot(filename, arrays, type(self).__name__, blob)
            return
        np.savez(filename, __name__=type(self).__name__, __args_and_kwargs__=np.void(blob),
                 __variables__=np.array(list(arrays.keys())), **{'var%03d' % i: v for i, v in enumerate(arrays.values())})

    @staticmethod
    def _read_snapshot(filename):
        """-> (class name, (ob_shape, nact), kwargs, {variable name: array})"""
        if filename.endswith('.h5'):
            try:
                import h5py
            except ImportError:
                h5py = None
            if h5py is not None:
                with h5py.File(filename, 'r') as f:
                    blob = f.attrs['args_and_kwargs'].tobytes()
                    arrays = {}
                    f.visititems(lambda n, o: arrays.__setitem__(n, o[...]) if isinstance(o, h5py.Dataset) else None)
                    name = f.attrs['name']
            else:
                from . import h5lite
                name, blob, arrays = h5lite.read_snapshot(filename)
            args, kwargs = _loads_spaces(blob)
            ob, ac = args[0], args[1]     # the reference pickles gym spaces; ours are (shape, n)
            ob_shape = getattr(ob, 'shape', None) or (ob.low.shape if hasattr(ob, 'low') else ob)
            nact = getattr(ac, 'n', None) or (ac.low.shape[0] if hasattr(ac, 'low') else ac)
            return name, (tuple(ob_shape), int(nact)), kwargs, arrays
        with np.load(filename, allow_pickle=False) as f:
            (ob_shape, nact), kwargs = pickle.loads(f['__args_and_kwargs__'].tobytes())
            names = [str(n) for n in f['__variables__']]
            arrays = {n: f['var%03d' % i] for i, n in enumerate(names)}
            return str(f['__name__']), (tuple(ob_shape), int(nact)), kwargs, arrays

    @classmethod
    def Load(cls, filename, engine=None, extra_kwargs=None):
        """policies.py:59-67"""
        _, (ob_shape, nact), kwargs, arrays = cls._read_snapshot(filename)
        if extra_kwargs:
            kwargs.update(extra_kwargs)
        pol = cls(_Space(shape=ob_shape), _Space(n=nact), engine=engine, **kwargs)
        pol._set_from_arrays(arrays, exact=True)
        return pol

    def _set_from_arrays(self, arrays, exact):
        scope = type(self).__name__
        flat = self.get_trainable_flat() if not exact else np.zeros(self.num_params, np.float32)
        for name, (off, shape) in self.spec.items():
            a = np.asarray(arrays['%s/%s:0' % (scope, name)], np.float32)
            if exact:
                assert a.shape == shape, (name, a.shape, shape)
            else:   # policies.py:355-369: the loaded arrays may be smaller; they fill the leading sub-array
                assert a.ndim == len(shape) and all(x >= y for x, y in zip(shape, a.shape)), \
                    'This policy must have more weights than the policy to load'
            view = flat[off:off + int(np.prod(shape))].reshape(shape)
            view[tuple(slice(0, n) for n in a.shape)] = a
        self.set_trainable_flat(flat)

    def initialize_from(self, filename):
        """policies.py:345-372: weights from another policy of the same architecture (variable names), whose arrays may
        be smaller than this policy's."""
        _, _, _, arrays = self._read_snapshot(filename)
        mine = set(self.variable_arrays().keys())
        assert mine == set(arrays.keys()), 'Variable names do not match'
        self._set_from_arrays(arrays, exact=False)


class ESAtariPolicy(Policy):
    """policies.py:305-429: conv8s4x16 -> BN -> conv4s2x32 -> BN -> fc256 -> BN -> out, argmax; virtual batch norm."""
    kind = _lib.KIND_ES

    def set_ref_batch(self, ref_batch):  # policies.py:332-335
        ref = np.asarray(ref_batch)
        if ref.dtype != np.uint8:       # observations are exact multiples of 1/255 (atari_wrappers.py:183-186)
            ref = np.rint(ref.astype(np.float32) * 255.0).astype(np.uint8)
        self.ref_batch = ref
        if self.engine is not None:
            self.engine.set_ref_batch(ref)

    @property
    def needs_ref_batch(self):
        return True

    def reinitialize(self):
        raise NotImplementedError("ESAtariPolicy has no reinitialize ops in the reference (layers.* variables)")

    def initialize(self, seed=0):
        self.set_trainable_flat(xavier_flat(self.num_actions, seed))

    def extra_variable_arrays(self, scope):
        """BatchNorm*/moving_mean, moving_variance (policies.py:322-328: batch_norm(decay=0., updates_collections=None)
        overwrites them with the batch moments of every is_training pass, i.e. of the reference batch).  The engine
        computes those moments in its reference pass for the unperturbed theta; without an engine or a reference
        batch they are TF's initial values (zeros / ones)."""
        n = {'BatchNorm': 16, 'BatchNorm_1': 32, 'BatchNorm_2': 256}
        mom = None
        if self.engine is not None and getattr(self, 'ref_batch', None) is not None and hasattr(self.engine, 'get_bn_moments'):
            e = self.engine
            e.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
            e.ref_pass(1)
            mom = e.get_bn_moments(1)[0]
        out, off = OrderedDict(), 0
        for bn, c in n.items():
            out['%s/%s/moving_mean:0' % (scope, bn)] = mom[off:off + c].copy() if mom is not None else np.zeros(c, np.float32)
            out['%s/%s/moving_variance:0' % (scope, bn)] = mom[off + c:off + 2 * c].copy() if mom is not None else np.ones(c, np.float32)
            off += 2 * c
        return out


class GAAtariPolicy(Policy):
    """policies.py:433-513: same trunk without batch norm, U.conv/U.dense with column-normalising reinitialize."""
    kind = _lib.KIND_GA

    def __init__(self, ob_space, ac_space, nonlin_type="relu", ac_init_std=0.1, engine=None):
        if nonlin_type != "relu":
            raise NotImplementedError("the HIP engine implements nonlin_type='relu' (configurations/frostbite_ga.json)")
        if ac_init_std != 0.1:
            raise NotImplementedError("ac_init_std is fixed at the reference default 0.1 (policies.py:434)")
        super().__init__(ob_space, ac_space, engine=engine, nonlin_type=nonlin_type, ac_init_std=ac_init_std)

    @property
    def needs_ref_batch(self):
        return False

    def set_from_seeds(self, seeds, noise_stdev):
        """ga.py:256-264 / 151-158: theta = normc(noise[s0]) + noise_stdev * sum noise[s_k], built on device."""
        self._flat = self.engine.ga_rebuild(0, np.asarray(seeds, np.int64), noise_stdev)
        return self._flat


class _GymBox(object):
    """stand-in whose pickle is renamed to gym.spaces.box.Box (state: low, high)"""


class _GymDiscrete(object):
    """stand-in whose pickle is renamed to gym.spaces.discrete.Discrete (state: n)"""


def _dumps_spaces(ob_shape, nact, kwargs):
    """policies.py:56: `pickle.dumps((self.args, self.kwargs))` where args are the gym spaces the policy was built with.  A stock
    checkout's Policy.Load does `cls(*args, **kwargs)` and reads ob_space.shape / ac_space.n (policies.py:306-309, 434-438), so
    the pickle must name gym's classes -- whether or not gym is installed here, and without touching sys.modules (another thread
    may be importing or unpickling gym objects at this moment).  The two stand-in classes above are pickled under their own
    names and the GLOBAL opcodes are then renamed in the byte stream: protocol 2 writes a class reference as the text
    `c<module>\n<name>\n`, with no length field in front of it, so the replacement is exact.  The result is the pickle gym 0.9.4
    (requirements.txt:3) produces: Box state {low, high}, Discrete state {n}.
    The two bound arrays are written as `numpy.zeros(shape)` / `numpy.ones(shape)` calls rather than as 2 x 226 KB of buffer:
    smaller, and free of the numpy-version-specific module path (`numpy._core` vs `numpy.core`) of a pickled ndarray."""
    import io
    ob, ac = _GymBox(), _GymDiscrete()
    ob.low, ob.high = np.zeros(ob_shape), np.ones(ob_shape)      # float64, as box.py builds them from scalar bounds
    ac.n = int(nact)

    class _Pickler(pickle.Pickler):
        def reducer_override(self, obj):
            if obj is ob.low:
                return np.zeros, (tuple(ob_shape),)
            if obj is ob.high:
                return np.ones, (tuple(ob_shape),)
            return NotImplemented
    out = io.BytesIO()
    _Pickler(out, protocol=2).dump(((ob, ac), kwargs))
    blob = out.getvalue()
    here = __name__.encode()
    for mine, (mod, name) in ((b'_GymBox', (b'gym.spaces.box', b'Box')), (b'_GymDiscrete', (b'gym.spaces.discrete', b'Discrete'))):
        ref = b'c' + here + b'\n' + mine + b'\n'
        assert blob.count(ref) == 1, 'unexpected pickle layout'
        blob = blob.replace(ref, b'c' + mod + b'\n' + name + b'\n')
    return blob


def _loads_spaces(blob):
    """pickle.loads for a snapshot's 'args_and_kwargs' (policies.py:56: the policy's gym spaces + kwargs): where gym is not
    installed its space classes are stood in for by attribute bags, which is all Load needs (shape / n)."""
    import io

    class _Unpickler(pickle.Unpickler):
        def find_class(self, mod, name):
            try:
                return super().find_class(mod, name)
            except (ImportError, AttributeError):
                if mod.split('.')[0] != 'gym':
                    raise
                return type(name, (), {'__module__': mod})
    return _Unpickler(io.BytesIO(blob)).load()


def snapshot_extension():
    """'.h5' (es.py:288-291's snapshot_iter*.h5) when this machine can write HDF5 -- h5py, or libhdf5 through h5lite -- else '.npz'"""
    try:
        import h5py  # noqa: F401
        return '.h5'
    except ImportError:
        from . import h5lite
        return '.h5' if h5lite.available() else '.npz'
