"""Drop-in for es_distributed/es.py on the HIP engine: same module surface (Config / Task / Result,
SharedNoiseTable, run_master, run_worker), with the inside of the worker's rollout loop (es.py:411-426)
and the master's reduce/update block (es.py:274-301) replaced by whole-population device calls.

One GPU worker evaluates its whole shard of antithetic pairs in a single dne_es_eval and pushes ONE Result
holding n pairs (legal: the master concatenates Results, es.py:274-277).  With several GPUs the shards are
exchanged as fixed 32-byte records by one all-gather (RCCL through torch.distributed) and every rank runs
the identical reduce + optimizer step, so theta stays bit-identical without a gradient all-reduce.
"""
import logging
import time
from collections import namedtuple

import numpy as np

from . import _lib
from .dist import MasterClient, WorkerClient

logger = logging.getLogger(__name__)

# wire types, es.py:12-23
Config = namedtuple('Config', [
    'l2coeff', 'noise_stdev', 'episodes_per_batch', 'timesteps_per_batch',
    'calc_obstat_prob', 'eval_prob', 'snapshot_freq',
    'return_proc_mode', 'episode_cutoff_mode'
])
Task = namedtuple('Task', ['params', 'ob_mean', 'ob_std', 'ref_batch', 'timestep_limit'])
Result = namedtuple('Result', [
    'worker_id',
    'noise_inds_n', 'returns_n2', 'signreturns_n2', 'lengths_n2',
    'eval_return', 'eval_length',
    'ob_sum', 'ob_sumsq', 'ob_count'
])

# one (noise_idx, returns, lengths, aux) record per antithetic pair -- what travels between GPUs
RECORD = np.dtype([('noise_idx', '<i8'), ('ret', '<f4', (2,)), ('len', '<i4', (2,)), ('aux', '<f4', (2,))])
assert RECORD.itemsize == 32


class SharedNoiseTable(object):
    """es.py:51-67.  Same stream (RandomState(seed).randn, float64 -> float32), same get / sample_index; the
    table additionally lives as one device buffer once attach()ed to an engine."""

    def __init__(self, count=250000000, seed=123):
        logger.info('Sampling {} random numbers with seed {}'.format(count, seed))
        self.noise = np.empty(count, dtype=np.float32)
        rs = np.random.RandomState(seed)
        chunk = 1 << 24  # consecutive randn calls continue the same stream; avoids the 2 GB float64 transient
        for i in range(0, count, chunk):
            n = min(chunk, count - i)
            self.noise[i:i + n] = rs.randn(n)  # 64-bit to 32-bit conversion here
        logger.info('Sampled {} bytes'.format(self.noise.size * 4))
        self._engines = []

    def get(self, i, dim):
        return self.noise[i:i + dim]

    def sample_index(self, stream, dim):
        return stream.randint(0, len(self.noise) - dim + 1)

    def attach(self, engine):
        """Upload once per engine (1 GB host -> HBM)."""
        if not any(e is engine for e in self._engines):
            engine.noise_upload(self.noise)
            self._engines.append(engine)
        return engine


def get_ref_batch(env, batch_size=32, random_stream=None):
    """es.py:105-113.  action_space.sample() draws from random_stream (gym's own stream is unseeded, Q1)."""
    rs = random_stream if random_stream is not None else np.random.RandomState()
    ref_batch = []
    ob = env.reset()
    while len(ref_batch) < batch_size:
        ob, rew, done, info = env.step(rs.randint(env.action_space.n))
        ref_batch.append(ob)
        if done:
            ob = env.reset()
    return ref_batch


def parse_cutoff(mode):
    """es.py:169-186 -> (tslimit, incr_threshold, incr_ratio, tslimit_max, adaptive)"""
    if isinstance(mode, int):
        return mode, None, None, mode, False
    if mode.startswith('adaptive:'):
        _, args = mode.split(':')
        a0, a1, a2, a3 = args.split(',')
        return int(a0), float(a1), float(a2), float(a3), True
    if mode == 'env_default':
        return None, None, None, None, False
    raise NotImplementedError(mode)


# ---------------------------------------------------------------------------------------------- sharding
def shard_pairs(n_pairs, rank, world):
    """Global pair ids owned by `rank`: round-robin, so long and short episodes mix on every GPU."""
    return np.arange(rank, n_pairs, world, dtype=np.int64)


def generation_inputs(noise_len, num_params, n_pairs, generation, rank, world):
    """Seeded stand-ins for the reference's unseeded streams: the worker's index stream is
    RandomState(generation*world + rank) (es.py:372,412), per-episode env seeds are RandomState(1000 +
    generation) u32 draws indexed by global pair id."""
    mine = shard_pairs(n_pairs, rank, world)
    rs = np.random.RandomState(generation * world + rank)
    idx = np.array([rs.randint(0, noise_len - num_params + 1) for _ in range(len(mine))], dtype=np.int64)
    all_seeds = np.random.RandomState(1000 + generation).randint(0, 2 ** 32, size=2 * n_pairs, dtype=np.uint64).astype(np.uint32)
    seeds = np.stack([all_seeds[2 * mine], all_seeds[2 * mine + 1]], axis=1).reshape(-1)
    return mine, idx, seeds


def pack_records(idx, returns_n2, lengths_n2, aux_n2):
    rec = np.zeros(len(idx), RECORD)
    rec['noise_idx'], rec['ret'], rec['len'], rec['aux'] = idx, returns_n2, lengths_n2, aux_n2
    return rec


def allgather_records(rec, n_pairs, rank, world, device=None):
    """The one exchange step of a generation: every rank contributes its shard's 32-byte records and gets
    all N back in global pair order.  torch.distributed all_gather ('nccl' = RCCL over xGMI on GPU tensors,
    'gloo' on CPU for the tests); world == 1 is a no-op."""
    if world == 1:
        return rec
    import torch
    import torch.distributed as dist
    per = (n_pairs + world - 1) // world
    buf = np.zeros(per, RECORD)
    buf[:len(rec)] = rec
    t = torch.from_numpy(buf.view(np.uint8).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t)
    allrec = out.cpu().numpy().view(RECORD).reshape(world, per)
    full = np.zeros(n_pairs, RECORD)
    for r in range(world):
        ids = shard_pairs(n_pairs, r, world)
        full[ids] = allrec[r, :len(ids)]
    return full


def es_generation(engine, noise_len, config, n_pairs, generation, tslimit, optimizer, rank=0, world=1, device=None):
    """One ES generation on this rank's shard + the redundant update (es.py:411-426 + 274-301).
    optimizer: dict(type='adam'|'sgd', args=dict(stepsize=..., ...)) as in the experiment JSON.
    Returns (records[N], update_ratio)."""
    P = engine.P
    mine, idx, seeds = generation_inputs(noise_len, P, n_pairs, generation, rank, world)
    ret, sg, ln = engine.es_eval(idx, config.noise_stdev, tslimit, seeds)
    rec = allgather_records(pack_records(idx, ret, ln, sg), n_pairs, rank, world, device)
    a = optimizer['args']
    ratio = engine.es_update(rec['noise_idx'], rec['ret'], rec['aux'], config.return_proc_mode, optimizer['type'],
                             config.l2coeff, a['stepsize'],
                             a.get('beta1', 0.9) if optimizer['type'] == 'adam' else a.get('momentum', 0.9),
                             a.get('beta2', 0.999), a.get('epsilon', 1e-08))
    return rec, ratio


# ---------------------------------------------------------------------------------------------- drivers
def make_engine(exp, n_pairs, n_actions=18, device_id=0, ref_count=128, **kw):
    kind = {'ESAtariPolicy': _lib.KIND_ES, 'GAAtariPolicy': _lib.KIND_GA}[exp['policy']['type']]
    return _lib.Engine(kind, n_actions, max_members=max(2 * n_pairs, 2), ref_count=ref_count, device_id=device_id, **kw)


def setup(exp, engine=None, n_pairs=None, device_id=0):
    """es.py:125-138: (config, env, policy) on the HIP engine (no TF session exists)."""
    from . import policies
    config = Config(**exp['config'])
    if engine is None:
        engine = make_engine(exp, n_pairs or max(config.episodes_per_batch // 2, 1), device_id=device_id)
    env = policies.HipAtariEnv(engine)
    policy = getattr(policies, exp['policy']['type'])(env.observation_space, env.action_space, engine=engine,
                                                      **exp['policy']['args'])
    return config, env, policy


def run_master(master_redis_cfg, log_dir, exp, *, engine=None, noise=None, max_iters=None, seed=0):
    """es.py:141-353.  Same loop: declare task -> pop results until episodes_per_batch and timesteps_per_batch
    are met -> process returns -> aggregate -> optimizer step, with the reduce running on the device
    (dne_es_update).  max_iters (extension) lets tests stop the otherwise endless loop."""
    from . import tabular_logger as tlogger
    logger.info('run_master: {}'.format(locals()))
    tlogger.start(log_dir)
    config, env, policy = setup(exp, engine=engine)
    engine = policy.engine
    master = MasterClient(master_redis_cfg)
    if policy.get_trainable_flat().any() == False:  # noqa: E712  fresh policy: reference relies on TF's initialiser
        policy.initialize(seed) if hasattr(policy, 'initialize') else None
    noise = noise if noise is not None else SharedNoiseTable()
    noise.attach(engine)
    engine.optimizer_reset()
    opt = exp['optimizer']
    if policy.needs_ref_batch:
        ref_batch = get_ref_batch(env, batch_size=engine.ref_count, random_stream=np.random.RandomState(seed))
        policy.set_ref_batch(ref_batch)
    tslimit, incr_tslimit_threshold, tslimit_incr_ratio, tslimit_max, adaptive_tslimit = parse_cutoff(config.episode_cutoff_mode)
    episodes_so_far = timesteps_so_far = 0
    tstart = time.time()
    master.declare_experiment(exp)
    it = 0
    while max_iters is None or it < max_iters:
        it += 1
        step_tstart = time.time()
        theta = policy.get_trainable_flat()
        assert theta.dtype == np.float32
        curr_task_id = master.declare_task(Task(
            params=theta, ob_mean=None, ob_std=None,
            ref_batch=policy.ref_batch if policy.needs_ref_batch else None, timestep_limit=tslimit))
        tlogger.log('********** Iteration {} **********'.format(curr_task_id))
        curr_task_results, eval_rets, eval_lens, worker_ids = [], [], [], []
        num_results_skipped = num_episodes_popped = num_timesteps_popped = 0
        while num_episodes_popped < config.episodes_per_batch or num_timesteps_popped < config.timesteps_per_batch:
            task_id, result = master.pop_result()
            assert isinstance(task_id, int) and isinstance(result, Result)
            assert (result.eval_return is None) == (result.eval_length is None)
            worker_ids.append(result.worker_id)
            if result.eval_length is not None:
                episodes_so_far += 1
                timesteps_so_far += result.eval_length
                if task_id == curr_task_id:
                    eval_rets.append(result.eval_return)
                    eval_lens.append(result.eval_length)
            else:
                assert (result.noise_inds_n.ndim == 1 and
                        result.returns_n2.shape == result.lengths_n2.shape == (len(result.noise_inds_n), 2))
                assert result.returns_n2.dtype == np.float32
                if task_id == curr_task_id:
                    episodes_so_far += result.lengths_n2.size
                    timesteps_so_far += result.lengths_n2.sum()
                    curr_task_results.append(result)
                    num_episodes_popped += result.lengths_n2.size
                    num_timesteps_popped += result.lengths_n2.sum()
                else:
                    num_results_skipped += 1
        frac_results_skipped = num_results_skipped / (num_results_skipped + len(curr_task_results))
        noise_inds_n = np.concatenate([r.noise_inds_n for r in curr_task_results])
        returns_n2 = np.concatenate([r.returns_n2 for r in curr_task_results])
        lengths_n2 = np.concatenate([r.lengths_n2 for r in curr_task_results])
        signreturns_n2 = np.concatenate([r.signreturns_n2 for r in curr_task_results])
        assert noise_inds_n.shape[0] == returns_n2.shape[0] == lengths_n2.shape[0]
        a = opt['args']
        # es.py:281-301 on the device: process returns, sum_i w_i * noise[idx_i], g /= 2N, -g + l2*theta, step
        update_ratio = engine.es_update(noise_inds_n, returns_n2, signreturns_n2, config.return_proc_mode, opt['type'],
                                        config.l2coeff, a['stepsize'],
                                        a.get('beta1', 0.9) if opt['type'] == 'adam' else a.get('momentum', 0.9),
                                        a.get('beta2', 0.999), a.get('epsilon', 1e-08))
        if adaptive_tslimit and (lengths_n2 == tslimit).mean() >= incr_tslimit_threshold:
            tslimit = min(int(tslimit_incr_ratio * tslimit), tslimit_max)
        step_tend = time.time()
        tlogger.record_tabular("EpRewMean", returns_n2.mean())
        tlogger.record_tabular("EpRewStd", returns_n2.std())
        tlogger.record_tabular("EpLenMean", lengths_n2.mean())
        tlogger.record_tabular("EvalEpRewMean", np.nan if not eval_rets else np.mean(eval_rets))
        tlogger.record_tabular("EvalEpCount", len(eval_rets))
        tlogger.record_tabular("Norm", float(np.square(policy.get_trainable_flat()).sum()))
        tlogger.record_tabular("UpdateRatio", float(update_ratio))
        tlogger.record_tabular("EpisodesThisIter", lengths_n2.size)
        tlogger.record_tabular("EpisodesSoFar", episodes_so_far)
        tlogger.record_tabular("TimestepsThisIter", lengths_n2.sum())
        tlogger.record_tabular("TimestepsSoFar", timesteps_so_far)
        tlogger.record_tabular("UniqueWorkers", len(set(worker_ids)))
        tlogger.record_tabular("ResultsSkippedFrac", frac_results_skipped)
        tlogger.record_tabular("TimeElapsedThisIter", step_tend - step_tstart)
        tlogger.record_tabular("TimestepsPerSecondThisIter", lengths_n2.sum() / (step_tend - step_tstart))
        tlogger.record_tabular("TimeElapsed", step_tend - tstart)
        tlogger.dump_tabular()
        if config.snapshot_freq != 0 and curr_task_id % config.snapshot_freq == 0:
            import os.path as osp
            filename = osp.join(log_dir, 'snapshot_iter{:05d}_rew{}.npz'.format(
                curr_task_id, np.nan if not eval_rets else int(np.mean(eval_rets))))
            policy.save(filename)
            tlogger.log('Saved snapshot {}'.format(filename))
    return policy


def run_worker(master_redis_cfg, relay_redis_cfg, noise, *, min_task_runtime=.2, engine=None, max_tasks=None,
               seed=None, rank=0, world=1):
    """es.py:366-439 for one GPU: per task, evaluate this worker's whole shard of antithetic pairs in one
    device call and push one Result with n pairs (exactly episodes_per_batch/2 pairs per
    generation across the `world` GPU workers).  min_task_runtime is accepted for signature parity."""
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
    n_pairs = max(config.episodes_per_batch // 2, 1)
    done_tasks, last_task = 0, None
    while max_tasks is None or done_tasks < max_tasks:
        task_id, task_data = worker.get_current_task()
        if task_id == last_task:   # one shard per task: wait for the next declaration instead of re-evaluating
            time.sleep(0.001)
            continue
        last_task = task_id
        assert isinstance(task_id, int) and isinstance(task_data, Task)
        if policy.needs_ref_batch:
            policy.set_ref_batch(task_data.ref_batch)
        policy.set_trainable_flat(task_data.params)
        tslimit = task_data.timestep_limit
        tslimit = _lib.ENV_MAX_EPISODE_STEPS if tslimit is None else min(tslimit, _lib.ENV_MAX_EPISODE_STEPS)
        if rs.rand() < config.eval_prob:
            # es.py:388-405: noiseless weights, reported separately, never part of the update
            engine.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
            er, _, el = engine.eval_members(1, tslimit, rs.randint(0, 2 ** 32, size=1, dtype=np.uint64).astype(np.uint32))
            worker.push_result(task_id, Result(worker_id=worker_id, noise_inds_n=None, returns_n2=None,
                                               signreturns_n2=None, lengths_n2=None, eval_return=float(er[0]),
                                               eval_length=int(el[0]), ob_sum=None, ob_sumsq=None, ob_count=None))
        mine = shard_pairs(n_pairs, rank, world)
        noise_inds = np.array([noise.sample_index(rs, policy.num_params) for _ in range(len(mine))], dtype=np.int64)
        seeds = rs.randint(0, 2 ** 32, size=2 * len(mine), dtype=np.uint64).astype(np.uint32)
        returns, signreturns, lengths = engine.es_eval(noise_inds, config.noise_stdev, tslimit, seeds)
        worker.push_result(task_id, Result(
            worker_id=worker_id, noise_inds_n=noise_inds, returns_n2=returns, signreturns_n2=signreturns,
            lengths_n2=lengths, eval_return=None, eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0))
        done_tasks += 1
