"""Drop-in for es_distributed/ga.py (Deep GA) on the HIP engine.

Individuals are lists of noise indices (seed chains).  A GPU worker builds a whole batch of children
(random parent chain + one fresh index, ga.py:251-254), evaluates them in ONE dne_ga_eval -- parents are
materialised once into device slots (normc(noise[s0]) + sigma * sum noise[s_k], ga.py:256-264) and cached
across generations, the child's own mutation is applied on the fly by the forward kernels -- and pushes one
Result whose noise_inds_n is the list of chains.  The master keeps elites (old score, not re-evaluated, Q6)
+ children and truncates with the deterministic order (-return, arrival index) (ga.py:136-149, Q5).
"""
import logging
import time

import numpy as np

from . import _lib
from .dist import MasterClient, WorkerClient
from .es import Config, Result, SharedNoiseTable, collect_batch, log_generation, namedtuple, parse_cutoff  # noqa: F401

logger = logging.getLogger(__name__)

GATask = namedtuple('GATask', ['params', 'population', 'ob_mean', 'ob_std', 'timestep_limit'])  # ga.py:4


def setup(exp, engine=None, n_children=None, device_id=0):
    """ga.py:7-20"""
    from . import policies
    config = Config(**exp['config'])
    if engine is None:
        engine = _lib.Engine(_lib.KIND_GA, 18, max_members=max(n_children or config.episodes_per_batch, 1), device_id=device_id)
    env = policies.HipAtariEnv(engine)
    policy = getattr(policies, exp['policy']['type'])(env.observation_space, env.action_space, engine=engine,
                                                      **exp['policy']['args'])
    return config, env, policy


def truncate(engine, noise_inds, returns, population_size):
    """ga.py:145-147 with the engine's deterministic selection."""
    idx = engine.ga_select(np.asarray(returns, np.float32), population_size)
    return [noise_inds[i] for i in idx], np.asarray(returns, np.float32)[idx]


def run_master(master_redis_cfg, log_dir, exp, *, engine=None, noise=None, max_iters=None):
    """ga.py:33-206"""
    from . import tabular_logger as tlogger
    logger.info('run_master: {}'.format(locals()))
    tlogger.start(log_dir)
    config, env, policy = setup(exp, engine=engine)
    engine = policy.engine
    master = MasterClient(master_redis_cfg)
    noise = noise if noise is not None else SharedNoiseTable()
    noise.attach(engine)
    tslimit, incr_tslimit_threshold, tslimit_incr_ratio, _, adaptive_tslimit = parse_cutoff(config.episode_cutoff_mode)
    episodes_so_far = timesteps_so_far = 0
    tstart = time.time()
    master.declare_experiment(exp)
    population, population_score = [], np.array([], np.float32)
    population_size, num_elites = exp['population_size'], exp['num_elites']
    it = 0
    while max_iters is None or it < max_iters:
        it += 1
        step_tstart = time.time()
        theta = policy.get_trainable_flat()
        assert theta.dtype == np.float32
        curr_task_id = master.declare_task(GATask(params=theta, population=population, ob_mean=None, ob_std=None,
                                                  timestep_limit=tslimit))
        tlogger.log('********** Iteration {} **********'.format(curr_task_id))
        batch = collect_batch(master, config, curr_task_id, check_pairs=False)
        episodes_so_far += batch.all_episodes
        timesteps_so_far += batch.all_timesteps
        curr_task_results, eval_rets = batch.results, batch.eval_rets
        # ga.py:136-149: elites (old scores) + all children, keep the best population_size
        noise_inds_n = [list(c) for c in population[:num_elites]]
        returns_n2 = list(population_score[:num_elites])
        for r in curr_task_results:
            noise_inds_n.extend(list(c) for c in r.noise_inds_n)
            returns_n2.extend(r.returns_n2)
        returns_n2 = np.array(returns_n2, np.float32)
        lengths_n2 = np.concatenate([r.lengths_n2 for r in curr_task_results])
        population, population_score = truncate(engine, noise_inds_n, returns_n2, population_size)
        assert len(population) == population_size
        assert np.max(returns_n2) == population_score[0]
        logger.info('Elite: {} score: {}'.format(population[0], population_score[0]))
        policy.set_from_seeds(population[0], config.noise_stdev)   # ga.py:151-158
        if adaptive_tslimit and (lengths_n2 == tslimit).mean() >= incr_tslimit_threshold:
            tslimit = int(tslimit_incr_ratio * tslimit)
        dt = time.time() - step_tstart
        log_generation(tlogger, [
            ("EpRewMax", returns_n2.max()), ("EpRewMean", returns_n2.mean()), ("EpLenMean", lengths_n2.mean()),
            ("EpisodesThisIter", lengths_n2.size), ("EpisodesSoFar", episodes_so_far),
            ("TimestepsThisIter", lengths_n2.sum()), ("TimestepsSoFar", timesteps_so_far),
            ("TimeElapsedThisIter", dt), ("TimestepsPerSecondThisIter", lengths_n2.sum() / dt),
            ("TimeElapsed", time.time() - tstart)])
        if config.snapshot_freq != 0:
            import os.path as osp
            policy.save(osp.join(log_dir, 'snapshot_iter{:05d}_rew{}.npz'.format(
                curr_task_id, np.nan if not eval_rets else int(np.mean(eval_rets)))))
    return policy, population, population_score


def run_worker(master_redis_cfg, relay_redis_cfg, noise, *, min_task_runtime=.2, engine=None, max_tasks=None, seed=None):
    """ga.py:209-284 for one GPU: all episodes_per_batch children of a task in one device call."""
    logger.info('run_worker: {}'.format(locals()))
    assert isinstance(noise, SharedNoiseTable)
    worker = WorkerClient(master_redis_cfg, relay_redis_cfg)   # reference quirk Q3: arguments swapped in ga.py:212
    exp = worker.get_experiment()
    config, env, policy = setup(exp, engine=engine)
    engine = policy.engine
    noise.attach(engine)
    rs = np.random.RandomState(seed)
    worker_id = rs.randint(2 ** 31)
    assert policy.needs_ob_stat == (config.calc_obstat_prob != 0)
    n = max(config.episodes_per_batch, 1)
    done_tasks, last_task = 0, None
    while max_tasks is None or done_tasks < max_tasks:
        task_id, task_data = worker.get_current_task()
        if task_id == last_task:
            time.sleep(0.001)
            continue
        last_task = task_id
        assert isinstance(task_id, int) and isinstance(task_data, GATask)
        tslimit = task_data.timestep_limit
        tslimit = _lib.ENV_MAX_EPISODE_STEPS if tslimit is None else min(tslimit, _lib.ENV_MAX_EPISODE_STEPS)
        chains = []
        for _ in range(n):   # ga.py:251-254
            if len(task_data.population) > 0:
                seeds = list(task_data.population[rs.randint(len(task_data.population))]) + [noise.sample_index(rs, policy.num_params)]
            else:
                seeds = [noise.sample_index(rs, policy.num_params)]
            chains.append(seeds)
        env_seeds = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        returns, signreturns, lengths = engine.ga_eval(chains, config.noise_stdev, tslimit, env_seeds)
        worker.push_result(task_id, Result(
            worker_id=worker_id, noise_inds_n=chains, returns_n2=returns, signreturns_n2=signreturns,
            lengths_n2=lengths, eval_return=None, eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0))
        done_tasks += 1
