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
from .policies import snapshot_extension
from .es import Config, Result, SharedNoiseTable, TaskPacer, collect_batch, log_generation, namedtuple, parse_cutoff  # noqa: F401

logger = logging.getLogger(__name__)

from .compat import GATask  # ga.py:4  # noqa: E402


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
            policy.save(osp.join(log_dir, ('snapshot_iter{:05d}_rew{}' + snapshot_extension()).format(
                curr_task_id, np.nan if not eval_rets else int(np.mean(eval_rets)))))
    return policy, population, population_score


def shard_children(n_children, rank, world):
    """children of a generation owned by `rank`: round-robin like es.shard_pairs"""
    return np.arange(rank, n_children, world, dtype=np.int64)


def make_children(population, n, noise, rs, num_params):
    """ga.py:251-254: a random parent's chain + one fresh index (generation 0: a single fresh index)"""
    chains = []
    for _ in range(n):
        if len(population) > 0:
            seeds = list(population[rs.randint(len(population))]) + [noise.sample_index(rs, num_params)]
        else:
            seeds = [noise.sample_index(rs, num_params)]
        chains.append(seeds)
    return chains


def run_worker(master_redis_cfg, relay_redis_cfg, noise, *, min_task_runtime=.2, engine=None, max_tasks=None, seed=None,
               rank=0, world=1, reeval_after=1.0):
    """ga.py:209-284 for one GPU: this worker's share (children rank, rank + world, ...) of the episodes_per_batch children
    of a task in one device call.  The reference constructs WorkerClient(master_redis_cfg, relay_redis_cfg) -- its
    arguments swapped against the class's (relay, master) order (ga.py:212); with distinct master / relay
    configurations that attaches the worker to the wrong server, so the order is put right here."""
    logger.info('run_worker: {}'.format(locals()))
    assert isinstance(noise, SharedNoiseTable)
    worker = WorkerClient(relay_redis_cfg, master_redis_cfg)
    exp = worker.get_experiment()
    config, env, policy = setup(exp, engine=engine)
    engine = policy.engine
    noise.attach(engine)
    rs = np.random.RandomState(seed)
    worker_id = rs.randint(2 ** 31)
    assert policy.needs_ob_stat == (config.calc_obstat_prob != 0)
    n = len(shard_children(max(config.episodes_per_batch, 1), rank, world))
    pacer = TaskPacer(worker, max_tasks, reeval_after)
    while True:
        nxt = pacer.next_task()
        if nxt is None:
            break
        task_id, task_data = nxt
        assert isinstance(task_id, int) and isinstance(task_data, GATask)
        tslimit = task_data.timestep_limit
        tslimit = _lib.ENV_MAX_EPISODE_STEPS if tslimit is None else min(tslimit, _lib.ENV_MAX_EPISODE_STEPS)
        if rs.rand() < config.eval_prob:
            # ga.py:226-245: one episode of the current elite (task_data.params), no task timestep limit, reported apart.
            # (The reference unpacks policy.rollout's three return values into two names there and would raise; this is
            # the branch as intended -- SURVEY Q2.)
            policy.set_trainable_flat(task_data.params)
            engine.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
            er, _, el = engine.eval_members(1, _lib.ENV_MAX_EPISODE_STEPS, rs.randint(0, 2 ** 32, size=1, dtype=np.uint64).astype(np.uint32))
            worker.push_result(task_id, Result(worker_id=worker_id, noise_inds_n=None, returns_n2=None, signreturns_n2=None,
                                               lengths_n2=None, eval_return=float(er[0]), eval_length=int(el[0]),
                                               ob_sum=None, ob_sumsq=None, ob_count=None))
        chains = make_children(task_data.population, n, noise, rs, policy.num_params)
        env_seeds = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        returns, signreturns, lengths = engine.ga_eval(chains, config.noise_stdev, tslimit, env_seeds)
        worker.push_result(task_id, Result(
            worker_id=worker_id, noise_inds_n=chains, returns_n2=returns, signreturns_n2=signreturns,
            lengths_n2=lengths, eval_return=None, eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0))
        pacer.pushed(task_id)


# ---------------------------------------------------------------------------------------------- co-located GPUs
# (the GA counterpart of es.es_generation: no broker, every rank keeps the whole population and runs the identical
#  truncation, so the elites agree on every GPU without a second exchange)
CHILD_RECORD = np.dtype([('parent', '<i4'), ('len', '<i4'), ('seed', '<i8'), ('ret', '<f4'), ('aux', '<f4'), ('pad', '<i8')])
assert CHILD_RECORD.itemsize == 32


def ga_generation_inputs(noise_len, num_params, n_children, n_parents, generation, rank, world):
    """Seeded stand-ins for the worker's unseeded stream (ga.py:216,251-254): parent picks and fresh indices from
    RandomState(generation * world + rank), env seeds from RandomState(1000 + generation) by global child id."""
    mine = shard_children(n_children, rank, world)
    rs = np.random.RandomState(generation * world + rank)
    parent = rs.randint(0, max(n_parents, 1), size=len(mine)).astype(np.int32) if n_parents > 0 else np.full(len(mine), -1, np.int32)
    fresh = rs.randint(0, noise_len - num_params + 1, size=len(mine)).astype(np.int64)
    all_seeds = np.random.RandomState(1000 + generation).randint(0, 2 ** 32, size=n_children, dtype=np.uint64).astype(np.uint32)
    return mine, parent, fresh, all_seeds[mine]


def ga_generation(engine, noise_len, sigma, population, scores, n_children, population_size, num_elites, generation, tslimit,
                  rank=0, world=1, transport=None):
    """One Deep-GA generation on this rank's share of the children + the redundant truncation (ga.py:251-271 + 136-149).
    Exchange: 32-byte child records (parent index, fresh seed, return, length), all-gathered over RCCL
    (engine.comm_allgather) or by `transport` (gloo in the CPU tests).  Returns (population, scores, lengths)."""
    mine, parent, fresh, env_seeds = ga_generation_inputs(noise_len, engine.P, n_children, len(population), generation, rank, world)
    chains = [(list(population[p]) if p >= 0 else []) + [int(f)] for p, f in zip(parent, fresh)]
    ret, sg, ln = engine.ga_eval(chains, sigma, tslimit, env_seeds)
    rec = np.zeros(len(mine), CHILD_RECORD)
    rec['parent'], rec['seed'], rec['ret'], rec['len'], rec['aux'] = parent, fresh, ret, ln, sg
    if world > 1:
        per = (n_children + world - 1) // world
        buf = np.zeros(per, CHILD_RECORD)
        buf[:len(rec)] = rec
        gathered = (engine.comm_allgather(buf) if transport is None else transport(buf, world)).reshape(world, per)
        full = np.zeros(n_children, CHILD_RECORD)
        for r in range(world):
            ids = shard_children(n_children, r, world)
            full[ids] = gathered[r, :len(ids)]
        rec = full
    # ga.py:136-149: elites keep their old score and are not re-evaluated (Q6); children follow in global child order
    cand = [list(c) for c in population[:num_elites]]
    cand_ret = list(scores[:num_elites])
    for r in rec:
        cand.append((list(population[r['parent']]) if r['parent'] >= 0 else []) + [int(r['seed'])])
        cand_ret.append(r['ret'])
    new_pop, new_scores = truncate(engine, cand, np.array(cand_ret, np.float32), population_size)
    return new_pop, new_scores, rec['len']
