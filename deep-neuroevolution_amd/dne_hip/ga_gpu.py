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


# ---------------------------------------------------------------------------------------------- mutation-power schedules
class Schedule(object):
    """The three schedules of neuroevolution/helper.py:46-88 as one value object.  The experiment JSON names them by the
    reference's class names ('ConstantSchedule', 'LinearSchedule', 'ExponentialSchedule') with that class's keyword arguments;
    progress is looked up by `field` among the keywords of value() ('iteration', 'timesteps_so_far': ga.py:73-74).
    linear:       start + min(progress / span, 1) * (stop - start)
    exponential:  the same interpolation between log(start) and log(stop), exponentiated.  (The reference's
                  ExponentialSchedule.value calls `self.linear(**kwargs)` on an object without __call__ and raises; the
                  interpolation in log space is what its constructor sets up.)"""
    KINDS = {'ConstantSchedule': 'constant', 'LinearSchedule': 'linear', 'ExponentialSchedule': 'exponential'}

    def __init__(self, kind, start, stop=None, span=None, field=None):
        if kind not in ('constant', 'linear', 'exponential'):
            raise ValueError('unknown schedule kind {!r}'.format(kind))
        self.kind, self.start, self.stop, self.span, self.field = kind, start, stop, span, field

    @classmethod
    def from_config(cls, spec):
        """helper.py:84-88: a bare number is a constant; otherwise {'type': <reference class name>, ...its keyword arguments}"""
        if isinstance(spec, numbers.Number):
            return cls('constant', spec)
        kw = dict(spec)
        kind = cls.KINDS[kw.pop('type')]
        if kind == 'constant':
            return cls(kind, kw['value'])
        return cls(kind, kw['initial_p'], kw['final_p'], kw['schedule'], kw['field'])

    def value(self, **progress):
        if self.kind == 'constant':
            return self.start
        # the reference asserts here (helper.py:61); callers that catch its AssertionError keep working
        assert self.field in progress, 'schedule needs {!r}; value() was given {}'.format(self.field, sorted(progress))
        reached = min(float(progress[self.field]) / self.span, 1.0)
        if self.kind == 'linear':
            return self.start + reached * (self.stop - self.start)
        lo, hi = math.log(self.start), math.log(self.stop)
        return math.exp(lo + reached * (hi - lo))


make_schedule = Schedule.from_config


class _LegacySchedule(Schedule):
    """snapshot.pkl files written before the three schedule classes became one hold TrainingState.mutation_power as an instance of
    dne_hip.ga_gpu.ConstantSchedule / LinearSchedule / ExponentialSchedule with those classes' own attributes (helper.py:46-82's:
    _value -- or schedule, field, initial_p, final_p).  These names keep such files loadable: unpickling hands the old attribute
    dict to __setstate__, which maps it onto Schedule's."""
    KIND = None

    def __init__(self, *args, **kw):
        spec = dict(kw, type=type(self).__name__)
        if args:                                   # the old positional forms: (value) / keyword-only otherwise
            spec['value'] = args[0]
        fresh = Schedule.from_config(spec)
        self.__dict__.update(fresh.__dict__)

    def __setstate__(self, state):
        if 'kind' in state:                        # written after the merge
            self.__dict__.update(state)
        elif self.KIND == 'constant':
            Schedule.__init__(self, 'constant', state['_value'])
        else:                                      # the old ExponentialSchedule also carried a helper object (`linear`): not needed
            Schedule.__init__(self, self.KIND, state['initial_p'], state['final_p'], state['schedule'], state['field'])


class ConstantSchedule(_LegacySchedule):
    KIND = 'constant'


class LinearSchedule(_LegacySchedule):
    KIND = 'linear'


class ExponentialSchedule(_LegacySchedule):
    KIND = 'exponential'


# ---------------------------------------------------------------------------------------------- run state (what snapshot.pkl holds)
class Offspring(object):
    """One evaluated genome (ga.py:88-106): seeds = (idx0, (idx1, power1), ...), the rewards / lengths of its training
    episodes and -- once validated -- of its validation episodes."""

    def __init__(self, seeds, rewards, ep_len, validation_rewards=(), validation_ep_len=()):
        self.seeds = seeds
        self.rewards, self.ep_len = rewards, ep_len
        self.validation_rewards, self.validation_ep_len = list(validation_rewards), list(validation_ep_len)

    fitness = property(lambda self: np.mean(self.rewards))
    training_steps = property(lambda self: np.sum(self.ep_len))


class TrainingState(object):
    """Everything a run needs to continue after a restart (ga.py:40-76): counters, the sorted population, the elite, the best
    validated solution, the mutation-power schedule and the episode cutoff with its adaptive growth rule."""
    COUNTERS = ('num_frames', 'timesteps_so_far', 'time_elapsed', 'validation_timesteps_so_far', 'it')

    def __init__(self, exp):
        from .es import parse_cutoff
        for name in self.COUNTERS:
            setattr(self, name, 0)
        self.population, self.elite = [], None
        self.curr_solution, self.curr_solution_val, self.curr_solution_test = None, float('-inf'), float('-inf')
        self.mutation_power = Schedule.from_config(exp['mutation_power'])
        limit, grow_at, grow_by, limit_max, adaptive = parse_cutoff(exp['episode_cutoff_mode'])
        self.tslimit, self.adaptive_tslimit = limit, adaptive
        self.incr_tslimit_threshold, self.tslimit_incr_ratio = grow_at, grow_by
        if adaptive:
            self.tslimit_max = limit_max

    def sample(self, schedule):
        return schedule.value(iteration=self.it, timesteps_so_far=self.timesteps_so_far)

    def copy_population(self, filename):
        """Start from another run's population (exp['load_population'], ga.py:78-86).  Snapshots written before mutation powers
        were stored per seed hold bare indices: those mutations were made at power 0.005."""
        with open(filename, 'rb') as f:
            self.population = pickle.load(f).population
        for o in self.population:
            root, rest = o.seeds[0], o.seeds[1:]
            o.seeds = (root, ) + tuple(m if isinstance(m, tuple) else (m, 0.005) for m in rest)


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
