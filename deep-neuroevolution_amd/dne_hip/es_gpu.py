"""The ES loop of the reference's GPU tree (gpu_implementation/es.py) on the HIP engine: one process, no Redis -- TrainingState with
`snapshot.pkl` resume (es.py:40-83, 155-162, 278-283), antithetic offspring at a scheduled mutation power (es.py:175-187), the
adaptive episode cutoff (es.py:52-62, 273-276), the elite's test episodes before and after every update (es.py:190, 249) and the
reference's tabular keys (es.py:206-268).

Model: configurations/es_atari_config.json names `ModelVirtualBN` (models/batchnorm.py:52-123: conv 16 8x8/4, conv 32 4x4/2, fc 256 with
virtual batch norm from a reference batch).  The engine's ES network IS that architecture in the es_distributed parameterisation
(policies.py:319-330: conv biases and a learnt BN scale in the flat vector, which ModelVirtualBN fixes at 0 and 1); exp['model'] =
'ModelVirtualBN' selects it, theta starts from policies.xavier_flat instead of the GPU tree's scaled noise slice (the parity
target is the CPU path).  What is NOT built: ModelVirtualBN's own flat layout (1 008 450 parameters), `load_from` (ga_legacy genomes).

Where the arithmetic lives: ranks, sum_i w_i * noise[idx_i] / 2N, -g + l2coeff * theta and the optimizer step are dne_es_update on the device
(the same formulas as es_distributed: es.py:227-246 here = es_distributed/es.py:281-301).  The GPU tree's SGD keeps v = momentum * v + g
(neuroevolution/optimizers.py:49-51) where es_distributed keeps (1 - momentum) * g: with u = (1 - momentum) * v that is the engine's SGD at
stepsize / (1 - momentum) -- equal in real arithmetic, not bit for bit; Adam is the same formula in both trees.
Unseeded streams of the reference (np.random.RandomState() at es.py:151, the environments' seeds) are seeded here.
"""
import os
import pickle
import time

import numpy as np

from . import _lib
from .es import SharedNoiseTable, get_ref_batch, optimizer_args, parse_cutoff
from .ga_gpu import Offspring, Schedule

MODEL_KINDS = {'ModelVirtualBN': _lib.KIND_ES}   # neuroevolution/models/batchnorm.py:52 (exp['model'], es.py:144)


class TrainingState(object):
    """What snapshot.pkl holds (es.py:40-83): the counters, the episode cutoff with its growth rule, the mutation-power schedule, theta and
    the optimizer's state -- here as plain arrays fetched from / pushed to the device around a pickle."""
    COUNTERS = ('num_frames', 'timesteps_so_far', 'time_elapsed', 'validation_timesteps_so_far', 'it')

    def __init__(self, exp):
        for name in self.COUNTERS:
            setattr(self, name, 0)
        self.mutation_power = Schedule.from_config(exp['mutation_power'])
        limit, grow_at, grow_by, limit_max, adaptive = parse_cutoff(exp['episode_cutoff_mode'])
        self.tslimit, self.adaptive_tslimit = limit, adaptive
        self.incr_tslimit_threshold, self.tslimit_incr_ratio = grow_at, grow_by
        if adaptive:
            self.tslimit_max = limit_max
        self.theta = None
        self.optimizer = None          # (m, v, t) of the device optimizer, None before the first update
        self.stream = None             # the index / environment-seed stream's position (an extension: the reference's stream is unseeded,
                                       # es.py:151; with it a resumed run continues exactly where an uninterrupted one would be)

    def sample(self, schedule):
        return schedule.value(iteration=self.it, timesteps_so_far=self.timesteps_so_far)

    def pull(self, engine):
        self.theta = engine.get_theta()
        self.optimizer = engine.optimizer_get_state()

    def push(self, engine):
        engine.set_theta(self.theta)
        engine.optimizer_reset()
        if self.optimizer is not None and self.optimizer[2] > 0:
            engine.optimizer_set_state(*self.optimizer)


def engine_optimizer(opt):
    """(kind, stepsize, beta1-or-momentum, beta2, epsilon) for dne_es_update; the GPU tree's SGD mapped as the module docstring says"""
    step, first, beta2, eps = optimizer_args(opt)
    if opt['type'] == 'sgd':
        step = step / (1.0 - first)
    return opt['type'], step, first, beta2, eps


def _episodes_of_theta(engine, n, tslimit, rs):
    """n episodes of the unperturbed theta (monitor_eval_repeated([(theta, 0)], ...), es.py:190, 249): pairs at mutation power 0 --
    theta + 0 * eps twice, every episode under its own environment seed"""
    limit = _lib.ENV_MAX_EPISODE_STEPS if tslimit is None else min(int(tslimit), _lib.ENV_MAX_EPISODE_STEPS)
    rets, lens = [], []
    left = int(n)
    while left > 0:
        pairs = min((left + 1) // 2, engine.max_members // 2)
        seeds = rs.randint(0, 2 ** 32, size=2 * pairs, dtype=np.uint64).astype(np.uint32)
        r, _, l = engine.es_eval(np.zeros(pairs, np.int64), 0.0, limit, seeds)
        rets.append(np.asarray(r).reshape(-1)); lens.append(np.asarray(l).reshape(-1))
        left -= 2 * pairs
    return np.concatenate(rets)[:n], np.concatenate(lens)[:n]


def main(log_dir, engine=None, noise=None, seed=0, max_iters=None, ref_count=128, **exp):
    """gpu_implementation/es.py:139-288.  Returns the TrainingState (theta and optimizer state pulled from the device)."""
    from . import policies, tabular_logger as tlogger
    tlogger.start(log_dir)
    if 'load_from' in exp:
        raise NotImplementedError("load_from (es.py:164-171: a ga_legacy genome as the first theta) is not built")
    n_pairs = exp['population_size'] // 2
    if engine is None:
        engine = _lib.Engine(MODEL_KINDS[exp['model']], 18, max_members=2 * n_pairs, ref_count=ref_count)
    noise = noise if noise is not None else SharedNoiseTable()
    noise.attach(engine)
    rs = np.random.RandomState(seed)
    all_tstart = tstart = time.time()
    try:                                                            # es.py:155-162: resume
        with open(os.path.join(log_dir, 'snapshot.pkl'), 'rb') as file:
            state = pickle.load(file)
        tlogger.log("Loaded iteration {} from {}".format(state.it, log_dir))
    except FileNotFoundError:
        state = TrainingState(exp)
        state.theta = policies.xavier_flat(engine.n_actions, seed)   # es.py:173: state.initialize(rs, noise, worker.model)
    state.push(engine)
    env = policies.HipAtariEnv(engine, seed=seed)                   # ModelVirtualBN.requires_ref_batch (batchnorm.py:60-62)
    ref = np.stack(get_ref_batch(env, batch_size=engine.ref_count, random_stream=np.random.RandomState(seed)))
    engine.set_ref_batch(np.rint(ref * 255.0).astype(np.uint8))
    opt = engine_optimizer(exp['optimizer'])
    initial_performance, _ = _episodes_of_theta(engine, exp['num_test_episodes'], None, rs)   # es.py:190
    if getattr(state, 'stream', None) is not None:
        rs.set_state(state.stream)
    iters = 0
    while max_iters is None or iters < max_iters:
        iters += 1
        tstart_iteration = time.time()
        if state.timesteps_so_far >= exp['timesteps']:
            break
        # es.py:175-187: population_size // 2 indices, each evaluated at +power and -power; 5000 frames = tslimit * 4 (es.py:198)
        power = state.sample(state.mutation_power)
        idx = np.array([noise.sample_index(rs, engine.P) for _ in range(n_pairs)], np.int64)
        seeds = rs.randint(0, 2 ** 32, size=2 * n_pairs, dtype=np.uint64).astype(np.uint32)
        limit = _lib.ENV_MAX_EPISODE_STEPS if state.tslimit is None else min(int(state.tslimit), _lib.ENV_MAX_EPISODE_STEPS)
        rets, sgn, lens = engine.es_eval(idx, power, limit, seeds)
        results = [Offspring(int(i), [float(r[0]), float(r[1])], [int(l[0]), int(l[1])]) for i, r, l in zip(idx, rets, lens)]
        state.num_frames += int(np.sum(lens)) * 4
        state.it += 1
        rewards = np.array([b for a in results for b in a.rewards])
        timesteps_this_iter = int(sum(a.training_steps for a in results))
        state.timesteps_so_far += timesteps_this_iter
        if exp['return_proc_mode'] != 'centered_rank':
            raise NotImplementedError(exp['return_proc_mode'])      # es.py:231-234
        # es.py:227-246 on the device: centered ranks of the 2N returns, g = sum_i (r+ - r-) noise[idx_i] / 2N, step on -g + l2coeff * theta
        update_ratio = engine.es_update(idx, rets, sgn, 'centered_rank', opt[0], exp['l2coeff'], *opt[1:])
        time_elapsed_this_iter = time.time() - tstart_iteration
        state.time_elapsed += time_elapsed_this_iter
        test_evals, test_lens = _episodes_of_theta(engine, exp['num_test_episodes'], None, rs)   # es.py:249
        dt = time.time() - tstart_iteration
        for key, val in (('Iteration', state.it), ('MutationPower', power), ('TimestepLimitPerEpisode', state.tslimit),
                         ('PopulationEpRewMax', np.max(rewards)), ('PopulationEpRewMean', np.mean(rewards)),
                         ('PopulationEpRewMedian', np.median(rewards)), ('PopulationEpCount', len(rewards)),
                         ('PopulationTimesteps', timesteps_this_iter), ('UpdateRatio', float(update_ratio)),
                         ('TestRewMean', np.mean(test_evals)), ('TestRewMedian', np.median(test_evals)), ('TestEpCount', len(test_evals)),
                         ('TestEpLenSum', int(np.sum(test_lens))), ('InitialRewMax', np.max(initial_performance)),
                         ('InitialRewMean', np.mean(initial_performance)), ('InitialRewMedian', np.median(initial_performance)),
                         ('TimestepsThisIter', timesteps_this_iter), ('TimestepsPerSecondThisIter', timesteps_this_iter / dt),
                         ('TimestepsComputed', state.num_frames), ('TimestepsSoFar', state.timesteps_so_far),
                         ('TimeElapsedThisIter', time_elapsed_this_iter), ('TimeElapsedThisIterTotal', dt),
                         ('TimeElapsed', state.time_elapsed), ('TimeElapsedTotal', time.time() - all_tstart)):
            tlogger.record_tabular(key, val)
        tlogger.dump_tabular()
        fps = state.timesteps_so_far / (time.time() - tstart)
        tlogger.log('Timesteps Per Second: {:.0f}. Elapsed: {:.2f}h'.format(fps, (time.time() - all_tstart) / 3600))
        if state.adaptive_tslimit:                                  # es.py:273-276 (a pair's two lengths summed against the limit, as written there)
            if np.mean([a.training_steps >= state.tslimit for a in results]) > state.incr_tslimit_threshold:
                state.tslimit = min(state.tslimit * state.tslimit_incr_ratio, state.tslimit_max)
                tlogger.log('Increased threshold to {}'.format(state.tslimit))
        state.pull(engine)                                          # es.py:278-283
        state.stream = rs.get_state()
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, 'snapshot.pkl'), 'wb') as file:
            pickle.dump(state, file)
        if state.timesteps_so_far >= exp['timesteps']:
            break
    state.pull(engine)
    return state
