"""Drop-in for es_distributed/es_modified.py -- the ES variant that produces the data VINE visualises: every rollout is
made reproducible by a policy seed (policies.py:392-396: env.seed(policy_seed)), every Result carries behaviour
characterisations (the RAM states; the dumps use the final one, es_modified.py:176,197: `bc_vec[-1]`), and the master
writes per generation
    snapshots/snapshot_gen_NNNN/snapshot_parent_NNNN.{h5|npz}, ..._rb.p     (policy + reference batch, es_modified.py:149-153)
    snapshots/snapshot_gen_NNNN/snapshot_parent_NNNN.dat                     (the eval episode closest to the mean, :162-179)
    snapshots/snapshot_gen_NNNN/snapshot_offspring_NNNN.dat                  (one row per perturbed rollout, :181-199)
in the reference's row format: 128 RAM bytes, fitness, length, then (seed, noise_stdev) or (noise_idx, policy_seed, sign).

On the device the perturbed rollouts of a whole shard run in one dne_es_eval with the final RAM of every member kept
(engine created with bc_final_only); every + rollout of a worker iteration uses policy_seed_pos as its environment seed
and every - rollout policy_seed_neg, exactly like the reference's loop (es_modified.py:489-497).  The reference appends
a Result's bc_vectors after its rollout loop, i.e. only for the LAST pair of each Result (es_modified.py:504-510, an
indentation quirk); a GPU worker's single Result stands for many of the reference's, so it carries every pair's.
"""
import csv
import logging
import os
import pickle

import numpy as np

from . import _lib
from . import es as _es
from .compat import Config, ModifiedResult as Result, Task   # noqa: F401
from .dist import WorkerClient
from .policies import snapshot_extension
from .es import SharedNoiseTable, TaskPacer, shard_pairs   # noqa: F401

logger = logging.getLogger(__name__)


def make_engine(exp, n_pairs, n_actions=18, device_id=0, ref_count=128):
    return _lib.Engine(_lib.KIND_ES, n_actions, max_members=max(2 * n_pairs, 2), ref_count=ref_count, device_id=device_id,
                       record_bc=True, bc_final_only=True)


def master_extract_parent(eval_bc_vecs, eval_rets, iteration, policy, ref_batch, root="snapshots"):
    """es_modified.py:140-179"""
    path = os.path.join(root, "snapshot_gen_{:04}".format(int(iteration)))
    os.makedirs(path, exist_ok=True)
    try:
        import h5py  # noqa: F401
        policy.save(os.path.join(path, "snapshot_parent_{:04d}.h5".format(iteration)))
    except ImportError:
        policy.save(os.path.join(path, ("snapshot_parent_{:04d}" + snapshot_extension()).format(iteration)))
    with open(os.path.join(path, "snapshot_parent_{:04d}_rb.p".format(iteration)), "wb") as f:
        pickle.dump(ref_batch, f)
    if not eval_rets:
        return
    eval_rets_arr = np.array(eval_rets)
    target = int(np.mean(eval_rets_arr))
    bc_vec, fitness, length, seed, noise_stdev = eval_bc_vecs[int((np.abs(eval_rets_arr - target)).argmin())]
    with open(os.path.join(path, "snapshot_parent_{:04}.dat".format(int(iteration))), "w+") as file:
        csv.writer(file, delimiter=' ').writerow(np.hstack((bc_vec[-1], fitness, length, seed, noise_stdev)))


def master_extract_cloud(curr_task_results, iteration, root="snapshots"):
    """es_modified.py:181-199"""
    path = os.path.join(root, "snapshot_gen_{:04}".format(int(iteration)))
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "snapshot_offspring_{:04}.dat".format(int(iteration))), "w+") as file:
        writer = csv.writer(file, delimiter=' ')
        for result in curr_task_results:
            for bc_vec, fitness, length, noise_idx, policy_seed, sign in result.bc_vectors:
                writer.writerow(np.hstack((bc_vec[-1], fitness, length, noise_idx, policy_seed, sign)))


def run_master(master_redis_cfg, log_dir, exp, *, engine=None, noise=None, max_iters=None, seed=0, snapshot_root=None):
    """es_modified.py:202-413 = es.run_master + the two dumps per generation (es_modified.py:332-333)."""
    root = snapshot_root if snapshot_root is not None else "snapshots"

    def dumps(task_id, batch, policy):
        master_extract_parent([r.bc_vectors[0] for r in batch.eval_results], batch.eval_rets, task_id, policy, policy.ref_batch, root)
        master_extract_cloud(batch.results, task_id, root)

    return _es.run_master(master_redis_cfg, log_dir, exp, engine=engine, noise=noise, max_iters=max_iters, seed=seed,
                          result_type=Result, on_generation=dumps)


def run_worker(master_redis_cfg, relay_redis_cfg, noise, *, min_task_runtime=.2, engine=None, max_tasks=None, seed=None,
               rank=0, world=1, reeval_after=1.0):
    """es_modified.py:429-527 for one GPU."""
    assert isinstance(noise, SharedNoiseTable)
    worker = WorkerClient(relay_redis_cfg, master_redis_cfg)
    exp = worker.get_experiment()
    config = Config(**exp['config'])
    n_pairs = max(config.episodes_per_batch // 2, 1)
    if engine is None:
        engine = make_engine(exp, len(shard_pairs(n_pairs, rank, world)))
    _, env, policy = _es.setup(exp, engine=engine)
    noise.attach(engine)
    rs = np.random.RandomState(seed)
    worker_id = rs.randint(2 ** 31)
    pacer = TaskPacer(worker, max_tasks, reeval_after)
    while True:
        nxt = pacer.next_task()
        if nxt is None:
            break
        task_id, task_data = nxt
        assert isinstance(task_id, int) and isinstance(task_data, Task)
        policy.set_ref_batch(task_data.ref_batch)
        policy.set_trainable_flat(task_data.params)
        tslimit = task_data.timestep_limit
        tslimit = _lib.ENV_MAX_EPISODE_STEPS if tslimit is None else min(tslimit, _lib.ENV_MAX_EPISODE_STEPS)
        # es_modified.py:451-455: three policy seeds per worker iteration (the reference draws them from an unseeded
        # np.random; here from the worker's stream so that a seeded worker is reproducible)
        policy_seed_eval, policy_seed_pos, policy_seed_neg = (int(x) for x in rs.randint(2 ** 20, size=3))
        if rs.rand() < config.eval_prob:   # es_modified.py:458-480 (this variant's eval rollouts DO obey the task's limit)
            engine.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
            er, _, el, bc = engine.eval_members(1, tslimit, np.array([policy_seed_eval], np.uint32), want_bc=True)
            worker.push_result(task_id, Result(
                worker_id=worker_id, noise_inds_n=None, returns_n2=None, signreturns_n2=None, lengths_n2=None,
                eval_return=float(er[0]), eval_length=int(el[0]), ob_sum=None, ob_sumsq=None, ob_count=None,
                bc_vectors=[(bc[0][None], float(er[0]), int(el[0]), policy_seed_eval, config.noise_stdev)]))
        mine = shard_pairs(n_pairs, rank, world)
        noise_inds = np.sort(np.array([noise.sample_index(rs, policy.num_params) for _ in range(len(mine))], dtype=np.int64))   # table order: see es.generation_inputs
        seeds = np.tile(np.array([policy_seed_pos, policy_seed_neg], np.uint32), len(mine))
        returns, signreturns, lengths, bc = engine.es_eval(noise_inds, config.noise_stdev, tslimit, seeds, want_bc=True)
        bc_vectors = []
        for i in range(len(mine)):   # es_modified.py:504-510, for every pair (see the module docstring)
            bc_vectors.append((bc[2 * i][None], returns[i, 0], int(lengths[i, 0]), int(noise_inds[i]), policy_seed_pos, 1))
            bc_vectors.append((bc[2 * i + 1][None], returns[i, 1], int(lengths[i, 1]), int(noise_inds[i]), policy_seed_neg, -1))
        worker.push_result(task_id, Result(
            worker_id=worker_id, noise_inds_n=noise_inds, returns_n2=returns, signreturns_n2=signreturns,
            lengths_n2=lengths, eval_return=None, eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0,
            bc_vectors=bc_vectors))
        pacer.pushed(task_id)
