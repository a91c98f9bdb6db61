"""Drop-in for es_distributed/es.py on the HIP engine: same module surface (Config / Task / Result,
SharedNoiseTable, run_master, run_worker), with the inside of the worker's rollout loop (es.py:411-426)
and the master's reduce/update block (es.py:274-301) replaced by whole-population device calls.

One GPU worker evaluates its whole shard of antithetic pairs in a single dne_es_eval and pushes ONE Result
holding n pairs (legal: the master concatenates Results, es.py:274-277).  With several GPUs the shards are
exchanged as fixed 32-byte records by one all-gather (RCCL behind the C ABI: dne_allgather_results; torch.distributed
only as the gloo carrier of the CPU tests) and every rank runs the identical reduce + optimizer step, so theta stays
bit-identical without a gradient all-reduce.
"""
import logging
import time
from collections import namedtuple

import os

import numpy as np

from . import _lib
from .dist import MasterClient, WorkerClient
from .policies import snapshot_extension

logger = logging.getLogger(__name__)

# wire types, es.py:12-23 (compat.py names them so that their pickles are the reference's)
from .compat import Config, Result, Task   # noqa: E402,F401

# one (noise_idx, returns, lengths, sign-returns) record per antithetic pair -- what travels between GPUs
RECORD = _lib.RECORD


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
    if mode == 'env_default':   # es.py:184-185: no task limit; the environment's own TimeLimit applies (policies.py:383-385)
        return None, None, None, None, False
    raise NotImplementedError(mode)


# ---------------------------------------------------------------------------------------------- sharding
def shard_pairs(n_pairs, rank, world):
    """Global pair ids owned by `rank`: round-robin, so long and short episodes mix on every GPU."""
    return np.arange(rank, n_pairs, world, dtype=np.int64)


def shard_mode():
    """How the ranks of a multi-GPU launch draw their noise indices (DNE_SHARD): 'table' (default) = every rank draws from ITS OWN
    1/world stretch of the table; 'uniform' = every rank draws from the whole table like an es_distributed worker (es.py:412).
    One rank: the two are the same thing."""
    m = os.environ.get("DNE_SHARD", "table")
    if m not in ("table", "uniform"):
        raise ValueError("DNE_SHARD must be 'table' or 'uniform', not %r" % m)
    return m


def index_range(noise_len, num_params, rank, world, shard=None):
    """[lo, hi) of the legal slice starts rank `rank` draws from"""
    hi = noise_len - num_params + 1
    if world > 1 and (shard or shard_mode()) == "table":
        return hi * rank // world, hi * (rank + 1) // world
    return 0, hi


def generation_inputs(noise_len, num_params, n_pairs, generation, rank, world, shard=None):
    """Seeded stand-ins for the reference's unseeded streams: the worker's index stream is
    RandomState(generation*world + rank) (es.py:372,412), per-episode env seeds are RandomState(1000 +
    generation) u32 draws indexed by global pair id.
    shard (matters for world > 1 only; default shard_mode()): 'uniform' -- every rank draws over the whole table, as a reference worker
    does; 'table' -- rank r draws over the r-th of `world` equal stretches of the legal start positions.  Each draw is uniform over
    its stretch and the stretches tile the table, so the population is a stratified sample of the same uniform distribution (every
    slice of the table is as likely as before, the update es.py:274-301 builds from the records is indifferent to who drew what).
    What it buys on MI355X: a rank's pairs lie as DENSE in its stretch as the whole population does in the whole table (2500 pairs
    over 250 M floats: every table row under ~10 slices), so the table-ordered streaming kernels and the caches share rows on a rank's
    share as they do at one GPU instead of pulling 4 MB per pair and step from HBM -- measured on one GPU, a rank of 2 / 4 / 8:
    126.3 / 57.9 / 38.3 ms per generation against 135.7 / 74.3 / 49.3 (profiles/r06_shard_ab.jsonl)."""
    mine = shard_pairs(n_pairs, rank, world)
    rs = np.random.RandomState(generation * world + rank)
    lo, hi = index_range(noise_len, num_params, rank, world, shard)
    # one vectorised draw = the same values, in the same order, as len(mine) successive SharedNoiseTable.sample_index calls
    # (legacy RandomState.randint with fixed bounds; pinned by tests/test_host_cpu.py)
    idx = rs.randint(lo, hi, size=len(mine)).astype(np.int64)
    # ascending table order: which of a worker's draws is "pair k" is a label (the reference's master takes results in arrival
    # order, es.py:246-260), and neighbours in the list then read neighbouring -- largely the same -- rows of the noise table,
    # which the table-ordered fc kernel (k_fc_duo) and the Infinity Cache turn into fewer HBM bytes
    idx.sort()
    all_seeds = np.random.RandomState(1000 + generation).randint(0, 2 ** 32, size=2 * n_pairs, dtype=np.uint64).astype(np.uint32)
    seeds = np.stack([all_seeds[2 * mine], all_seeds[2 * mine + 1]], axis=1).reshape(-1)
    return mine, idx, seeds


def pack_records(idx, returns_n2, lengths_n2, aux_n2):
    rec = np.zeros(len(idx), RECORD)
    rec['noise_idx'], rec['ret'], rec['len'], rec['aux'] = idx, returns_n2, lengths_n2, aux_n2
    return rec


def allgather_records(rec, n_pairs, rank, world, device=None):
    """Host transport of the exchange step (the CPU tests' gloo path, or any torch.distributed backend a deployment
    already has): every rank contributes its shard's 32-byte records and gets all N back in global pair order.
    On MI355X nodes the exchange is Engine.allgather_results instead -- RCCL behind the C ABI, device to device."""
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


def es_generation(engine, noise_len, config, n_pairs, generation, tslimit, optimizer, rank=0, world=1, transport=None):
    """One ES generation on this rank's shard + the redundant update (es.py:411-426 + 274-301).
    optimizer: dict(type='adam'|'sgd', args=dict(stepsize=..., ...)) as in the experiment JSON.
    The records never leave the device on the default path: dne_allgather_results packs them from the evaluation's
    accumulators, all-gathers them over RCCL (engine.comm_init beforehand when world > 1) and dne_es_update_gathered
    consumes the gathered buffer.  transport(rec, n_pairs, rank, world) -> all records is the host alternative
    (allgather_records over gloo in the CPU tests).  Returns (records[N], update_ratio)."""
    P = engine.P
    mine, idx, seeds = generation_inputs(noise_len, P, n_pairs, generation, rank, world)
    engine.es_eval(idx, config.noise_stdev, tslimit, seeds)
    if transport is None:
        rec = engine.allgather_results(len(mine), n_pairs)
    else:
        rec = transport(engine.records_pack(len(mine)), n_pairs, rank, world)
        engine.records_set(rec)
    ratio = engine.es_update_gathered(config.return_proc_mode, optimizer['type'], config.l2coeff, *optimizer_args(optimizer))
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


class Batch:
    """What the master keeps of one generation: the Results that belong to the current task, eval episodes,
    and the bookkeeping counters the reference logs (es.py:226-277)."""

    def __init__(self):
        self.results, self.eval_rets, self.eval_lens, self.worker_ids, self.eval_results = [], [], [], [], []
        self.skipped = self.episodes = self.timesteps = 0
        self.all_episodes = self.all_timesteps = 0   # including stale tasks and eval jobs (EpisodesSoFar / TimestepsSoFar)

    def cat(self, field):
        return np.concatenate([getattr(r, field) for r in self.results])

    @property
    def skipped_frac(self):
        return self.skipped / max(self.skipped + len(self.results), 1)


def collect_batch(master, config, task_id, check_pairs=True, result_type=None):
    """Pop Results until both episodes_per_batch and timesteps_per_batch are met (es.py:230-265): results of
    older tasks are counted but dropped, eval jobs are kept apart, shapes/dtypes asserted like the reference."""
    b = Batch()
    while b.episodes < config.episodes_per_batch or b.timesteps < config.timesteps_per_batch:
        tid, res = master.pop_result()
        assert isinstance(tid, int) and isinstance(res, result_type or Result)
        assert (res.eval_return is None) == (res.eval_length is None)
        b.worker_ids.append(res.worker_id)
        if res.eval_length is not None:                   # an evaluation episode of the unperturbed theta
            b.all_episodes += 1
            b.all_timesteps += res.eval_length
            if tid == task_id:
                b.eval_rets.append(res.eval_return)
                b.eval_lens.append(res.eval_length)
                b.eval_results.append(res)
            continue
        if check_pairs:                                   # es.py:246-248
            assert res.noise_inds_n.ndim == 1
            assert res.returns_n2.shape == res.lengths_n2.shape == (len(res.noise_inds_n), 2)
        assert res.returns_n2.dtype == np.float32
        if tid != task_id:
            b.skipped += 1
            continue
        n_eps, n_steps = res.lengths_n2.size, int(res.lengths_n2.sum())
        b.results.append(res)
        b.episodes += n_eps
        b.timesteps += n_steps
        b.all_episodes += n_eps
        b.all_timesteps += n_steps
    return b


def optimizer_args(opt):
    """(stepsize, beta1-or-momentum, beta2, epsilon) from the experiment's optimizer dict (optimizers.py:24,36)."""
    a = opt['args']
    first = a.get('beta1', 0.9) if opt['type'] == 'adam' else a.get('momentum', 0.9)
    return a['stepsize'], first, a.get('beta2', 0.999), a.get('epsilon', 1e-08)


def log_generation(tlogger, rows):
    for k, v in rows:
        tlogger.record_tabular(k, v)
    tlogger.dump_tabular()


def run_master(master_redis_cfg, log_dir, exp, *, engine=None, noise=None, max_iters=None, seed=0, result_type=None,
               on_generation=None):
    """es.py:141-353.  Same protocol: declare a task -> collect Results -> process returns -> aggregate ->
    optimizer step, with the reduce running on the device (dne_es_update).  max_iters (extension) lets tests
    stop the otherwise endless loop.  result_type / on_generation(task_id, batch, policy) are the seams es_modified.py
    uses (a Result with bc_vectors; the per-generation dumps)."""
    from . import tabular_logger as tlogger
    logger.info('run_master: {}'.format(locals()))
    tlogger.start(log_dir)
    config, env, policy = setup(exp, engine=engine)
    engine = policy.engine
    master = MasterClient(master_redis_cfg)
    if not policy.get_trainable_flat().any() and hasattr(policy, 'initialize'):
        policy.initialize(seed)                           # the reference relies on TF's (unseeded) initialiser
    noise = noise if noise is not None else SharedNoiseTable()
    noise.attach(engine)
    engine.optimizer_reset()
    opt = exp['optimizer']
    if policy.needs_ref_batch:                            # es.py:160-162
        policy.set_ref_batch(get_ref_batch(env, batch_size=engine.ref_count, random_stream=np.random.RandomState(seed)))
    tslimit, grow_thresh, grow_ratio, tslimit_max, adaptive = parse_cutoff(config.episode_cutoff_mode)
    episodes_so_far = timesteps_so_far = 0
    tstart = time.time()
    master.declare_experiment(exp)
    it = 0
    while max_iters is None or it < max_iters:
        it += 1
        t0 = time.time()
        theta = policy.get_trainable_flat()
        assert theta.dtype == np.float32
        task_id = master.declare_task(Task(params=theta, ob_mean=None, ob_std=None, timestep_limit=tslimit,
                                           ref_batch=policy.ref_batch if policy.needs_ref_batch else None))
        tlogger.log('********** Iteration {} **********'.format(task_id))
        batch = collect_batch(master, config, task_id, result_type=result_type)
        if on_generation is not None:      # es_modified.py:332-333 dumps before the update
            on_generation(task_id, batch, policy)
        episodes_so_far += batch.all_episodes
        timesteps_so_far += batch.all_timesteps
        noise_inds_n, returns_n2 = batch.cat('noise_inds_n'), batch.cat('returns_n2')
        lengths_n2, signreturns_n2 = batch.cat('lengths_n2'), batch.cat('signreturns_n2')
        assert noise_inds_n.shape[0] == returns_n2.shape[0] == lengths_n2.shape[0]
        # es.py:281-301 on the device: process returns, sum_i w_i * noise[idx_i], g /= 2N, -g + l2*theta, step
        update_ratio = engine.es_update(noise_inds_n, returns_n2, signreturns_n2, config.return_proc_mode, opt['type'],
                                        config.l2coeff, *optimizer_args(opt))
        if adaptive and (lengths_n2 == tslimit).mean() >= grow_thresh:      # es.py:308-311
            tslimit = min(int(grow_ratio * tslimit), tslimit_max)
        dt = time.time() - t0
        ev = batch.eval_rets
        log_generation(tlogger, [
            ("EpRewMean", returns_n2.mean()), ("EpRewStd", returns_n2.std()), ("EpLenMean", lengths_n2.mean()),
            ("EvalEpRewMean", np.mean(ev) if ev else np.nan), ("EvalEpCount", len(ev)),
            ("Norm", float(np.square(policy.get_trainable_flat()).sum())), ("UpdateRatio", float(update_ratio)),
            ("EpisodesThisIter", lengths_n2.size), ("EpisodesSoFar", episodes_so_far),
            ("TimestepsThisIter", lengths_n2.sum()), ("TimestepsSoFar", timesteps_so_far),
            ("UniqueWorkers", len(set(batch.worker_ids))), ("ResultsSkippedFrac", batch.skipped_frac),
            ("TimeElapsedThisIter", dt), ("TimestepsPerSecondThisIter", lengths_n2.sum() / dt),
            ("TimeElapsed", time.time() - tstart)])
        if config.snapshot_freq != 0 and task_id % config.snapshot_freq == 0:    # es.py:345-353
            import os.path as osp
            fn = osp.join(log_dir, ('snapshot_iter{:05d}_rew{}' + snapshot_extension()).format(task_id, int(np.mean(ev)) if ev else np.nan))
            policy.save(fn)
            tlogger.log('Saved snapshot {}'.format(fn))
    return policy


class TaskPacer:
    """Worker-side pacing.  The reference's CPU workers loop without pause, a few episodes per Result (es.py:377-439),
    so the master's collection loop (es.py:230-265: until episodes_per_batch AND timesteps_per_batch) always ends.  A
    GPU worker delivers its whole shard in one Result; if the master has still not declared a new task `reeval_after`
    seconds later it evidently needs more (odd episodes_per_batch, short episodes vs. timesteps_per_batch), and the
    worker evaluates another shard with fresh indices instead of sleeping forever.  max_tasks (tests) counts distinct
    task ids that received at least one Result."""

    def __init__(self, worker, max_tasks=None, reeval_after=1.0):
        self.worker, self.max_tasks, self.reeval_after = worker, max_tasks, reeval_after
        self.last_task, self.pushed_at, self.distinct = None, 0.0, 0

    def next_task(self):
        """-> (task_id, task_data), or None when max_tasks distinct tasks have been served"""
        while True:
            task_id, task_data = self.worker.get_current_task()
            if task_id != self.last_task:
                if self.max_tasks is not None and self.distinct >= self.max_tasks:
                    return None
                return task_id, task_data
            if self.max_tasks is not None and self.distinct >= self.max_tasks:
                return None
            if time.time() - self.pushed_at >= self.reeval_after:
                return task_id, task_data
            time.sleep(0.001)

    def pushed(self, task_id):
        if task_id != self.last_task:
            self.distinct += 1
        self.last_task, self.pushed_at = task_id, time.time()


def run_worker(master_redis_cfg, relay_redis_cfg, noise, *, min_task_runtime=.2, engine=None, max_tasks=None,
               seed=None, rank=0, world=1, reeval_after=1.0):
    """es.py:366-439 for one GPU: per task, evaluate this worker's whole shard of antithetic pairs in one
    device call and push one Result with n pairs (exactly episodes_per_batch/2 pairs per
    generation across the `world` GPU workers).  min_task_runtime is accepted for signature parity.
    Deviation, on purpose: a reference worker's loop iteration is EITHER an evaluation episode (probability eval_prob,
    es.py:388-405) OR a batch of pairs; a CPU fleet of hundreds of workers delivers both kinds all the time.  One GPU worker
    delivers the whole generation in a single Result, so an iteration that only evaluated would leave the master without its
    batch until the pacer re-evaluates: here the coin adds the evaluation episode (pushed first, as its own Result, never part
    of the update) and the shard is evaluated in the same iteration."""
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
    pacer = TaskPacer(worker, max_tasks, reeval_after)
    while True:
        nxt = pacer.next_task()
        if nxt is None:
            break
        task_id, task_data = nxt
        assert isinstance(task_id, int) and isinstance(task_data, Task)
        if policy.needs_ref_batch:
            policy.set_ref_batch(task_data.ref_batch)
        policy.set_trainable_flat(task_data.params)
        tslimit = task_data.timestep_limit
        tslimit = _lib.ENV_MAX_EPISODE_STEPS if tslimit is None else min(tslimit, _lib.ENV_MAX_EPISODE_STEPS)
        if rs.rand() < config.eval_prob:
            # es.py:388-405: noiseless weights, reported separately, never part of the update; "eval rollouts don't obey
            # task_data.timestep_limit" (es.py:391) -- only the environment's own limit applies
            engine.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
            er, _, el = engine.eval_members(1, _lib.ENV_MAX_EPISODE_STEPS, rs.randint(0, 2 ** 32, size=1, dtype=np.uint64).astype(np.uint32))
            worker.push_result(task_id, Result(worker_id=worker_id, noise_inds_n=None, returns_n2=None,
                                               signreturns_n2=None, lengths_n2=None, eval_return=float(er[0]),
                                               eval_length=int(el[0]), ob_sum=None, ob_sumsq=None, ob_count=None))
        mine = shard_pairs(n_pairs, rank, world)
        noise_inds = np.sort(np.array([noise.sample_index(rs, policy.num_params) for _ in range(len(mine))], dtype=np.int64))   # table order: see generation_inputs
        seeds = rs.randint(0, 2 ** 32, size=2 * len(mine), dtype=np.uint64).astype(np.uint32)
        returns, signreturns, lengths = engine.es_eval(noise_inds, config.noise_stdev, tslimit, seeds)
        worker.push_result(task_id, Result(
            worker_id=worker_id, noise_inds_n=noise_inds, returns_n2=returns, signreturns_n2=signreturns,
            lengths_n2=lengths, eval_return=None, eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0))
        pacer.pushed(task_id)
