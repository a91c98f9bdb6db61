"""The Deep-GA loop of the reference's GPU tree (gpu_implementation/ga.py) on the HIP engine: mutation-power schedules
(neuroevolution/helper.py:46-88), TrainingState / Offspring with `snapshot.pkl` resume (ga.py:40-112, 135-143, 251-256),
genomes ((idx0,), (idx1, power1), ...) with a scaled-noise root (models/base.py:118-149), truncation selection with a
validated elite (ga.py:166-204, 263-274).

Models: `Model` (models/dqn.py:24-37: conv 16 8x8/4, conv 32 4x4/2, fc 256, out) -- its forward is the engine's GAAtariPolicy
network -- and `LargeModel` (dqn.py:39-47: conv 32/64/64, fc 512, the one configurations/ga_atari_config.json names), the engine's
DNE_KIND_GA_LARGE (csrc/forward_large.h); exp['model'] picks one like ga.py:110.
The reference evaluates through TensorFlow workers (ConcurrentWorkers.monitor_eval); here a generation is one
dne_ga_eval_powers call and the validation / test episodes are batched calls of the same entry point.
Unseeded streams of the reference (np.random.RandomState() in ga.py:127, the environments' own seeds) are seeded here.
"""
import math
import numbers
import os
import pickle
import time

import numpy as np

from . import _lib
from .policies import flat_layout


# ---------------------------------------------------------------------------------------------- helper.py:46-88
class ConstantSchedule(object):
    def __init__(self, value):
        self._value = value

    def value(self, **kwargs):
        return self._value


class LinearSchedule(object):
    def __init__(self, schedule, final_p, initial_p, field):
        self.schedule, self.field, self.final_p, self.initial_p = schedule, field, final_p, initial_p

    def value(self, **kwargs):
        assert self.field in kwargs, "Argument {} not provided to scheduler Available: {}".format(self.field, kwargs)
        fraction = min(float(kwargs[self.field]) / self.schedule, 1.0)
        return self.initial_p + fraction * (self.final_p - self.initial_p)


class ExponentialSchedule(object):
    """helper.py:69-82 interpolates linearly in log space.  (The reference's `value` calls `self.linear(**kwargs)` on an object
    that has no __call__ and would raise; `.value(**kwargs)` is what it means.)"""

    def __init__(self, initial_p, final_p, schedule, field):
        self.initial_p, self.final_p, self.schedule, self.field = initial_p, final_p, schedule, field
        self.linear = LinearSchedule(initial_p=math.log(initial_p), final_p=math.log(final_p), schedule=schedule, field=field)

    def value(self, **kwargs):
        return math.exp(self.linear.value(**kwargs))


def make_schedule(args):
    """helper.py:84-88: a number -> ConstantSchedule, else {'type': <class name>, ...kwargs}"""
    if isinstance(args, numbers.Number):
        return ConstantSchedule(args)
    return {'ConstantSchedule': ConstantSchedule, 'LinearSchedule': LinearSchedule,
            'ExponentialSchedule': ExponentialSchedule}[args['type']](**{k: v for k, v in args.items() if k != 'type'})


# ---------------------------------------------------------------------------------------------- ga.py:40-112
class TrainingState(object):
    def __init__(self, exp):
        self.num_frames = 0
        self.population = []
        self.timesteps_so_far = 0
        self.time_elapsed = 0
        self.validation_timesteps_so_far = 0
        self.elite = None
        self.it = 0
        self.mutation_power = make_schedule(exp['mutation_power'])
        self.curr_solution = None
        self.curr_solution_val = float('-inf')
        self.curr_solution_test = float('-inf')
        mode = exp['episode_cutoff_mode']
        if isinstance(mode, int):
            self.tslimit, self.incr_tslimit_threshold, self.tslimit_incr_ratio, self.adaptive_tslimit = mode, None, None, False
        elif mode.startswith('adaptive:'):
            a0, a1, a2, a3 = mode.split(':')[1].split(',')
            self.tslimit, self.incr_tslimit_threshold, self.tslimit_incr_ratio, self.tslimit_max = int(a0), float(a1), float(a2), float(a3)
            self.adaptive_tslimit = True
        elif mode == 'env_default':
            self.tslimit, self.incr_tslimit_threshold, self.tslimit_incr_ratio, self.adaptive_tslimit = None, None, None, False
        else:
            raise NotImplementedError(mode)

    def sample(self, schedule):
        return schedule.value(iteration=self.it, timesteps_so_far=self.timesteps_so_far)

    def copy_population(self, filename):
        """ga.py:78-86 incl. the back-compatibility rule: bare seeds get the mutation power 0.005"""
        with open(filename, 'rb+') as file:
            state = pickle.load(file)
            self.population = state.population
            for offspring in self.population:
                offspring.seeds = (offspring.seeds[0], ) + tuple(s if isinstance(s, tuple) else (s, 0.005) for s in offspring.seeds[1:])


class Offspring(object):
    def __init__(self, seeds, rewards, ep_len, validation_rewards=[], validation_ep_len=[]):
        self.seeds, self.rewards, self.ep_len = seeds, rewards, ep_len
        self.validation_rewards, self.validation_ep_len = validation_rewards, validation_ep_len

    @property
    def fitness(self):
        return np.mean(self.rewards)

    @property
    def training_steps(self):
        return np.sum(self.ep_len)


# ---------------------------------------------------------------------------------------------- models/dqn.py:24-37, base.py:190-201
def model_scale_by(nact, kind=None):
    """scale_by of `Model` / `LargeModel` (dqn.py:25-27, inherited by LargeModel): weights std / sqrt(prod(shape[:-1])) with std 1.0
    (out layer 0.1), biases 0 -- in the engine's flat order = the creation order of dqn.py:29-36 / 39-47."""
    spec, P = flat_layout(_lib.KIND_GA if kind is None else kind, nact)
    sb = np.zeros(P, np.float32)
    for name, (off, shape) in spec.items():
        n = int(np.prod(shape))
        if name.endswith('/w'):
            std = 0.1 if name.startswith('out') else 1.0
            sb[off:off + n] = np.float32(std / np.sqrt(np.prod(shape[:-1])))
    return sb


MODEL_KINDS = {'Model': _lib.KIND_GA, 'LargeModel': _lib.KIND_GA_LARGE}   # neuroevolution/models/dqn.py:24-47 (exp['model'], ga.py:110)


class HipModel(object):
    """randomize / mutate / compute_weights_from_seeds of models/base.py:113-149, with theta living on the device"""

    def __init__(self, engine):
        self.engine, self.num_params = engine, engine.P
        self.scale_by = model_scale_by(engine.n_actions, engine.kind)
        engine.ga_set_init_scale(self.scale_by)

    def randomize(self, rs, noise):
        return (noise.sample_index(rs, self.num_params), )

    def mutate(self, parent_seeds, rs, noise, mutation_power):
        return tuple(parent_seeds) + ((noise.sample_index(rs, self.num_params), mutation_power), )

    def compute_weights_from_seeds(self, noise, seeds, cache=None):
        return self.engine.ga_rebuild_powers(0, seeds)


def _evaluate(engine, genomes, tslimit, rs):
    """(returns, lengths) of one episode per genome; env seeds from `rs` (the reference's environments are unseeded)"""
    limit = _lib.ENV_MAX_EPISODE_STEPS if tslimit is None else min(int(tslimit), _lib.ENV_MAX_EPISODE_STEPS)
    out_r, out_l = [], []
    for s in range(0, len(genomes), engine.max_members):
        part = genomes[s:s + engine.max_members]
        ret, _, ln = engine.ga_eval_powers(part, limit, rs.randint(0, 2 ** 32, size=len(part), dtype=np.uint64).astype(np.uint32))
        out_r.append(ret); out_l.append(ln)
    return np.concatenate(out_r), np.concatenate(out_l)


def main(log_dir, engine=None, noise=None, seed=0, max_iters=None, **exp):
    """gpu_implementation/ga.py:114-275.  Returns (curr_solution_test, {'val': curr_solution_val}, state)."""
    from . import tabular_logger as tlogger
    from .es import SharedNoiseTable
    tlogger.start(log_dir)
    if engine is None:
        engine = _lib.Engine(MODEL_KINDS[exp.get('model', 'Model')], 18, max_members=exp['population_size'])
    noise = noise if noise is not None else SharedNoiseTable()
    noise.attach(engine)
    model = HipModel(engine)
    rs = np.random.RandomState(seed)
    all_tstart = time.time()
    try:                                                           # ga.py:135-143: resume
        with open(os.path.join(log_dir, 'snapshot.pkl'), 'rb+') as file:
            state = pickle.load(file)
        tlogger.log("Loaded iteration {} from {}".format(state.it, log_dir))
    except FileNotFoundError:
        state = TrainingState(exp)
    if 'load_population' in exp:
        state.copy_population(exp['load_population'])

    def parents_of(state):                                         # ga.py:147-156, 263-274: the elite first, then the top selection_threshold
        T = exp['selection_threshold']
        if not state.population or T <= 0:
            return []
        top = [o.seeds for o in state.population[:T]]
        if state.elite is None or state.elite.seeds in top:
            return top
        return [state.elite.seeds] + top[:T - 1]

    cached_parents = parents_of(state)
    iters = 0
    while max_iters is None or iters < max_iters:
        iters += 1
        tstart_iteration = time.time()
        if state.timesteps_so_far >= exp['timesteps']:
            break
        assert (len(cached_parents) == 0 and state.it == 0) or len(cached_parents) == exp['selection_threshold']
        power = state.sample(state.mutation_power)
        tasks = [model.randomize(rs, noise) if not cached_parents else
                 model.mutate(cached_parents[rs.randint(len(cached_parents))], rs, noise, mutation_power=power)
                 for _ in range(exp['population_size'])]                                       # ga.py:128-133, 161
        rets, lens = _evaluate(engine, tasks, state.tslimit, rs)
        results = [Offspring(s, [float(r)], [int(l)]) for s, r, l in zip(tasks, rets, lens)]    # ga.py:162-163
        state.num_frames += int(lens.sum()) * 4
        state.it += 1
        rewards = np.array([a.fitness for a in results])
        population_timesteps = sum(a.training_steps for a in results)
        # ga.py:176: sorted(..., reverse=True) is stable, so equal fitness keeps arrival order -- the engine's selection
        # order (-fitness, arrival index)
        order = engine.ga_select(rewards.astype(np.float32), len(results))
        state.population = [results[i] for i in order]
        validation_population = state.population[:exp['validation_threshold']]                # ga.py:184-186
        if state.elite is not None:
            validation_population = [state.elite] + validation_population[:-1]
        vt = [o.seeds for o in validation_population for _ in range(exp['num_validation_episodes'])]
        vr, vl = _evaluate(engine, vt, state.tslimit, rs)                                      # ga.py:188-192
        k = exp['num_validation_episodes']
        population_validation = [float(np.mean(vr[i * k:(i + 1) * k])) for i in range(len(validation_population))]
        population_validation_len = [int(np.sum(vl[i * k:(i + 1) * k])) for i in range(len(validation_population))]
        state.elite = validation_population[int(np.argmax(population_validation))]            # ga.py:198-199
        er, el = _evaluate(engine, [state.elite.seeds] * exp['num_test_episodes'], None, rs)   # ga.py:200-201 (max_frames=None)
        validation_timesteps = sum(population_validation_len)
        timesteps_this_iter = population_timesteps + validation_timesteps
        state.timesteps_so_far += timesteps_this_iter
        state.validation_timesteps_so_far += validation_timesteps
        if np.mean(population_validation) > state.curr_solution_val:                           # ga.py:223-226
            state.curr_solution = state.elite.seeds
            state.curr_solution_val = float(np.mean(population_validation))
            state.curr_solution_test = float(np.mean(er))
        dt = time.time() - tstart_iteration
        state.time_elapsed += dt
        for key, val in (('Iteration', state.it), ('MutationPower', power), ('PopulationEpRewMax', np.max(rewards)),
                         ('PopulationEpRewMean', np.mean(rewards)), ('PopulationEpCount', len(rewards)),
                         ('PopulationTimesteps', population_timesteps), ('NumSelectedIndividuals', exp['selection_threshold']),
                         ('TruncatedPopulationRewMean', np.mean([a.fitness for a in validation_population])),
                         ('TruncatedPopulationValidationRewMean', np.mean(population_validation)),
                         ('TruncatedPopulationEliteValidationRewMean', np.max(population_validation)),
                         ('TruncatedPopulationEliteTestRewMean', np.mean(er)), ('TruncatedPopulationEliteTestEpCount', len(er)),
                         ('TruncatedPopulationEliteTestEpLenSum', int(np.sum(el))), ('ValidationTimestepsThisIter', validation_timesteps),
                         ('TimestepsThisIter', timesteps_this_iter), ('TimestepsPerSecondThisIter', timesteps_this_iter / dt),
                         ('TimestepsSoFar', state.timesteps_so_far), ('TimeElapsedThisIter', dt), ('TimeElapsed', state.time_elapsed),
                         ('TimeElapsedTotal', time.time() - all_tstart)):
            tlogger.record_tabular(key, val)
        tlogger.dump_tabular()
        if state.adaptive_tslimit:                                                              # ga.py:244-247
            if np.mean([a.training_steps >= state.tslimit for a in results]) > state.incr_tslimit_threshold:
                state.tslimit = min(state.tslimit * state.tslimit_incr_ratio, state.tslimit_max)
        os.makedirs(log_dir, exist_ok=True)                                                     # ga.py:249-254
        with open(os.path.join(log_dir, 'snapshot.pkl'), 'wb+') as file:
            pickle.dump(state, file)
        if state.timesteps_so_far >= exp['timesteps']:
            break
        cached_parents = parents_of(state)                                                      # ga.py:261-274
    return float(state.curr_solution_test), {'val': float(state.curr_solution_val)}, state
