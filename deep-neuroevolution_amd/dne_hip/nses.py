"""Drop-in for es_distributed/nses.py (NS-ES / NSR-ES) on the HIP engine.

Same meta-population loop (nses.py:58-316): pop_size parameter vectors, each with its own optimizer, an
archive of behaviour characterisations (BC = the RAM trajectory of one unperturbed rollout), novelty = mean
distance to the k nearest archive entries (nses.py:12-32).  On the device: the perturbed rollouts record their
RAM trajectories in HBM (they never cross PCIe), dne_novelty_batch scores all 2N of them against the archive,
and the novelty travels in the Result's signreturns slot exactly like the reference (nses.py:384, Q7).
"""
import logging
import time

import numpy as np

from . import _lib
from .dist import MasterClient, WorkerClient
from .es import (Config, Result, SharedNoiseTable, Task, TaskPacer, collect_batch, get_ref_batch, log_generation,
                 optimizer_args, parse_cutoff, shard_pairs)

logger = logging.getLogger(__name__)


def compute_novelty_vs_archive(engine, archive, novelty_vector, k):
    """nses.py:22-32 (one trajectory, host buffers)"""
    return engine.novelty(archive, novelty_vector, k)


def get_mean_bc(engine, tslimit, seed, num_rollouts=1):
    """nses.py:34-39 for the unperturbed theta in slot 0.  num_rollouts must be 1 (Q9: the reference's mean over
    ragged trajectories is only defined for one rollout; configurations/frostbite_nses.json uses 1)."""
    assert num_rollouts == 1
    engine.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
    _, _, ln, bc = engine.eval_members(1, tslimit, np.array([seed], np.uint32), want_bc=True)
    return bc[0, :ln[0]].copy()


def bc_capacity(tslimit_max):
    """Steps of RAM trajectory to keep per member.  episode_cutoff_mode 'env_default' has no task limit (es.py:184-185):
    the environment's own limit (400k raw frames = 100k skip-4 steps) bounds the trajectory."""
    return int(tslimit_max) if tslimit_max is not None else _lib.ENV_MAX_EPISODE_STEPS // 4


def make_engine(exp, n_pairs, tslimit_max, n_actions=18, device_id=0, ref_count=128):
    tslimit_max = bc_capacity(tslimit_max)
    return _lib.Engine(_lib.KIND_ES, n_actions, max_members=max(2 * n_pairs, 2), ref_count=ref_count, device_id=device_id,
                       record_bc=True, bc_max_steps=int(tslimit_max))


def run_master(master_redis_cfg, log_dir, exp, *, engine=None, noise=None, max_iters=None, seed=0):
    from . import policies, tabular_logger as tlogger
    logger.info('run_master: {}'.format(locals()))
    tlogger.start(log_dir)
    config = Config(**exp['config'])
    algo_type = exp['algo_type']
    tslimit, incr_tslimit_threshold, tslimit_incr_ratio, tslimit_max, adaptive_tslimit = parse_cutoff(config.episode_cutoff_mode)
    if engine is None:
        engine = make_engine(exp, max(config.episodes_per_batch // 2, 1), tslimit_max)
    env = policies.HipAtariEnv(engine)
    policy = policies.ESAtariPolicy(env.observation_space, env.action_space, engine=engine, **exp['policy']['args'])
    master = MasterClient(master_redis_cfg)
    noise = noise if noise is not None else SharedNoiseTable()
    noise.attach(engine)
    rs = np.random.RandomState(seed)
    ref_batch = get_ref_batch(env, batch_size=engine.ref_count, random_stream=np.random.RandomState(seed))
    policy.set_ref_batch(ref_batch)
    ns = exp['novelty_search']
    pop_size, num_rollouts = int(ns['population_size']), int(ns['num_rollouts'])
    opt = exp['optimizer']
    theta_dict, optimizer_dict = {}, {}
    curr_parent = 0
    P = policy.num_params
    for p in range(pop_size):   # nses.py:95-117
        theta = policies.xavier_flat(policy.num_actions, seed + p)
        policy.set_trainable_flat(theta)
        master.add_to_novelty_archive(get_mean_bc(engine, bc_capacity(tslimit_max), rs.randint(2 ** 31), num_rollouts))
        theta_dict[p] = theta
        optimizer_dict[p] = (np.zeros(P, np.float32), np.zeros(P, np.float32), 0)
    master.declare_experiment(exp)
    tstart = time.time()
    it = 0
    while max_iters is None or it < max_iters:
        it += 1
        step_tstart = time.time()
        theta = theta_dict[curr_parent]
        policy.set_trainable_flat(theta)
        engine.optimizer_set_state(*optimizer_dict[curr_parent])
        curr_task_id = master.declare_task(Task(params=theta, ob_mean=None, ob_std=None, ref_batch=policy.ref_batch,
                                                timestep_limit=tslimit))
        tlogger.log('********** Iteration {} **********'.format(curr_task_id))
        batch = collect_batch(master, config, curr_task_id)
        noise_inds_n, returns_n2 = batch.cat('noise_inds_n'), batch.cat('returns_n2')
        lengths_n2, signreturns_n2 = batch.cat('lengths_n2'), batch.cat('signreturns_n2')   # novelty rides in signreturns (Q7)
        # nses.py:217-228
        if config.return_proc_mode == 'centered_rank':
            proc = engine.centered_ranks(returns_n2)
        elif config.return_proc_mode == 'sign':
            proc = signreturns_n2
        elif config.return_proc_mode == 'centered_sign_rank':
            proc = engine.centered_ranks(signreturns_n2)
        else:
            raise NotImplementedError(config.return_proc_mode)
        if algo_type == "nsr":
            rew_ranks = engine.centered_ranks(returns_n2)
            proc = ((rew_ranks + proc) / 2.0).astype(np.float32)
        engine.weighted_sum(noise_inds_n, proc[:, 0] - proc[:, 1], float(returns_n2.size), copy_out=False)   # nses.py:231-236
        update_ratio = engine.optimizer_step(opt['type'], config.l2coeff, *optimizer_args(opt))
        mean_bc = get_mean_bc(engine, bc_capacity(tslimit_max), rs.randint(2 ** 31), num_rollouts)   # nses.py:246-247
        master.add_to_novelty_archive(mean_bc)
        if adaptive_tslimit and (lengths_n2 == tslimit).mean() >= incr_tslimit_threshold:
            tslimit = min(int(tslimit_incr_ratio * tslimit), int(tslimit_max))
        dt = time.time() - step_tstart
        log_generation(tlogger, [
            ("ParentId", curr_parent), ("EpRewMean", returns_n2.mean()), ("EpLenMean", lengths_n2.mean()),
            ("NoveltyMean", signreturns_n2.mean()), ("UpdateRatio", float(update_ratio)),
            ("TimestepsThisIter", lengths_n2.sum()), ("TimestepsPerSecondThisIter", lengths_n2.sum() / dt),
            ("TimeElapsed", time.time() - tstart)])
        theta_dict[curr_parent] = policy.get_trainable_flat()       # nses.py:283-284
        optimizer_dict[curr_parent] = engine.optimizer_get_state()
        if ns['selection_method'] == "novelty_prob":                  # nses.py:293-302
            archive = master.get_archive()
            novelty_probs = []
            for p in range(pop_size):
                policy.set_trainable_flat(theta_dict[p])
                bc = get_mean_bc(engine, bc_capacity(tslimit_max), rs.randint(2 ** 31), num_rollouts)
                novelty_probs.append(compute_novelty_vs_archive(engine, archive, bc, ns['k']))
            novelty_probs = np.array(novelty_probs) / float(np.sum(novelty_probs))
            curr_parent = rs.choice(range(pop_size), 1, p=novelty_probs)[0]
        elif ns['selection_method'] == "round_robin":
            curr_parent = (curr_parent + 1) % pop_size
        else:
            raise NotImplementedError(ns['selection_method'])
    return theta_dict, master.get_archive()


def run_worker(master_redis_cfg, relay_redis_cfg, noise, *, min_task_runtime=.2, engine=None, max_tasks=None, seed=None,
               rank=0, world=1, reeval_after=1.0):
    """nses.py:318-400 for one GPU (see es.run_worker); the novelty of both rollouts of every pair is computed on
    the device from the recorded RAM trajectories and shipped in signreturns_n2."""
    from . import policies
    assert isinstance(noise, SharedNoiseTable)
    worker = WorkerClient(relay_redis_cfg, master_redis_cfg)
    exp = worker.get_experiment()
    config = Config(**exp['config'])
    _, _, _, tslimit_max, _ = parse_cutoff(config.episode_cutoff_mode)
    n_pairs = max(config.episodes_per_batch // 2, 1)
    if engine is None:
        engine = make_engine(exp, n_pairs, tslimit_max)
    env = policies.HipAtariEnv(engine)
    policy = policies.ESAtariPolicy(env.observation_space, env.action_space, engine=engine, **exp['policy']['args'])
    noise.attach(engine)
    rs = np.random.RandomState(seed)
    worker_id = rs.randint(2 ** 31)
    k = exp['novelty_search']['k']
    pacer = TaskPacer(worker, max_tasks, reeval_after)
    cap = bc_capacity(tslimit_max)
    while True:
        nxt = pacer.next_task()
        if nxt is None:
            break
        task_id, task_data = nxt
        archive = worker.get_archive()                                 # nses.py:342-344
        policy.set_ref_batch(task_data.ref_batch)
        policy.set_trainable_flat(task_data.params)
        tslimit = cap if task_data.timestep_limit is None else min(task_data.timestep_limit, cap)
        mine = shard_pairs(n_pairs, rank, world)
        noise_inds = np.sort(np.array([noise.sample_index(rs, policy.num_params) for _ in range(len(mine))], dtype=np.int64))   # table order: see generation_inputs
        seeds = rs.randint(0, 2 ** 32, size=2 * len(mine), dtype=np.uint64).astype(np.uint32)
        returns, _, lengths = engine.es_eval(noise_inds, config.noise_stdev, tslimit, seeds)
        novelty = engine.novelty_batch(archive, lengths, k).astype(np.float32).reshape(-1, 2)   # nses.py:381-384
        worker.push_result(task_id, Result(
            worker_id=worker_id, noise_inds_n=noise_inds, returns_n2=returns, signreturns_n2=novelty,
            lengths_n2=lengths, eval_return=None, eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0))
        pacer.pushed(task_id)


# ---------------------------------------------------------------------------------------------- co-located GPUs (config 4)
def blend_and_update(engine, rec, algo_type, return_proc_mode, l2coeff, optimizer):
    """nses.py:217-236 on the gathered records: process the novelty (riding in the aux slot, Q7) by return_proc_mode, for
    NSR-ES average with the reward ranks, then aggregate and step.  Identical inputs on every rank -> identical theta."""
    returns_n2, nov_n2 = np.ascontiguousarray(rec['ret']), np.ascontiguousarray(rec['aux'])
    if return_proc_mode == 'centered_rank':
        proc = engine.centered_ranks(returns_n2)
    elif return_proc_mode == 'sign':
        proc = nov_n2
    elif return_proc_mode == 'centered_sign_rank':
        proc = engine.centered_ranks(nov_n2)
    else:
        raise NotImplementedError(return_proc_mode)
    if algo_type == "nsr":
        proc = ((engine.centered_ranks(returns_n2) + proc) / 2.0).astype(np.float32)
    engine.weighted_sum(rec['noise_idx'], proc[:, 0] - proc[:, 1], float(returns_n2.size), copy_out=False)
    return engine.optimizer_step(optimizer['type'], l2coeff, *optimizer_args(optimizer))


def nses_generation(engine, noise_len, config, algo_type, archive, k, n_pairs, generation, tslimit, optimizer, rank=0, world=1,
                    transport=None):
    """One NS-ES / NSR-ES generation of the current parent on this rank's shard: rollouts record their
    RAM trajectories in HBM, dne_novelty_batch scores them against the (replicated, device-resident) archive, only the
    32-byte records -- novelty in the aux slot -- travel (engine.comm_allgather = RCCL, or `transport` on the host), and
    every rank runs the same blend + update.  Returns (records[N], update_ratio)."""
    from .es import RECORD, generation_inputs, pack_records
    mine, idx, seeds = generation_inputs(noise_len, engine.P, n_pairs, generation, rank, world)
    ret, _, ln = engine.es_eval(idx, config.noise_stdev, tslimit, seeds)
    nov = engine.novelty_batch(archive, ln, k).astype(np.float32).reshape(-1, 2)       # nses.py:381-384
    rec = pack_records(idx, ret, ln, nov)
    if world > 1:
        per = (n_pairs + world - 1) // world
        buf = np.zeros(per, RECORD)
        buf[:len(rec)] = rec
        gathered = (engine.comm_allgather(buf) if transport is None else transport(buf, world)).reshape(world, per)
        full = np.zeros(n_pairs, RECORD)
        for r in range(world):
            ids = shard_pairs(n_pairs, r, world)
            full[ids] = gathered[r, :len(ids)]
        rec = full
    ratio = blend_and_update(engine, rec, algo_type, config.return_proc_mode, config.l2coeff, optimizer)
    return rec, ratio
