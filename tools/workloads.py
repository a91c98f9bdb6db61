#!/usr/bin/env python3
"""The BASELINE.json configurations next to the headline one, as functions: bench.py times them for its `extra` block and the
tools/*.py command lines print them.

  config 1  frostbite_es.json schema, pop 256, sigma 0.02, exactly 2 CPU worker processes (the reference CPU path; oracle)
  config 3  Deep GA, 1000 children per generation, top-20 parents, 1 elite -- es_distributed genomes on the small network
            (ga.py:136-149, 251-271) and the GPU tree's protocol on its LargeModel (configurations/ga_atari_config.json)
  config 4  NS-ES pop 5000: meta-population of 3, k = 10, behaviour characterisation = RAM trajectory, novelty on the device
            against a replicated archive (nses.py:217-228, 293-302, 381-384)
  config 5  the six-game ES loop (gym_tensorflow/atari/tf_atari.py:158-212: Asteroids has 14 actions, the others 18)

Every function takes (rank, world): the population is sharded exactly like the headline workload and the extra engine
borrows the headline engine's RCCL communicator (dne_comm_share), or exchanges through `transport` on the host.
ALE and the ROMs do not exist in this image: every game is the SynthAtari fixture with that game's action count.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "deep-neuroevolution_amd"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK = 8.0e12
BYTES_PER_STEP = {   # every weight of a member once per env-step + the u8 observation stack
    "es": 4 * 1009058 + 28224, "ga": 4 * 1008450 + 28224, "ga_large": 4 * 4052658 + 28224}
GAMES = {"frostbite": 18, "seaquest": 18, "asteroids": 14, "gravitar": 18, "venture": 18, "zaxxon": 18}   # tf_atari.py:158
ES_CONFIG = {"calc_obstat_prob": 0.0, "episodes_per_batch": 5000, "eval_prob": 0.0, "l2coeff": 0.005, "noise_stdev": 0.02,
             "snapshot_freq": 0, "timesteps_per_batch": 10000, "return_proc_mode": "centered_rank", "episode_cutoff_mode": 5000}
ES_OPT = {"args": {"stepsize": 0.01}, "type": "adam"}


def whole_job_roofline(steps_per_s, kind, world=1):
    """achieved = env-steps/s x algorithmic bytes per env-step of that network (whole job: reset, rebuild, tail, exchange and
    selection included in the time), against the 8 TB/s spec of each GPU used"""
    b = BYTES_PER_STEP[kind]
    return {"bound": "hbm", "achieved": steps_per_s * b / 1e9, "peak": HBM_PEAK * world / 1e9, "unit": "GB/s",
            "frac": steps_per_s * b / (HBM_PEAK * world), "traffic": None, "algorithmic_bytes_per_unit": b,
            "unit_def": "one env-step of one member", "basis": "whole job (wall time of the generation)"}


def _share(engine, comm_from, world, transport):
    if world > 1 and transport is None:
        if comm_from is None:
            raise RuntimeError("world > 1 without a transport needs the engine that owns the RCCL communicator")
        engine.comm_share(comm_from)


def _es_engine(noise, nact, n_pairs, rank, world, device_id, theta_seed, env_seed, ref_seed, comm_from, transport, **kw):
    from dne_hip import _lib, es, policies
    mine = len(es.shard_pairs(n_pairs, rank, world))
    e = _lib.Engine(_lib.KIND_ES, nact, max_members=max(2 * mine, 2), ref_count=128, device_id=device_id, **kw)
    _share(e, comm_from, world, transport)
    noise.attach(e)
    e.set_theta(policies.xavier_flat(nact, seed=theta_seed))
    env = policies.HipAtariEnv(e, seed=env_seed)
    ref = np.rint(np.stack(es.get_ref_batch(env, 128, np.random.RandomState(ref_seed))) * 255.0).astype(np.uint8)
    e.set_ref_batch(ref)
    e._bench_ref = ref          # the CPU legs time the oracle on the same reference batch
    e.optimizer_reset()
    return e


# ------------------------------------------------------------------------------------------------ config 3
def ga_small(noise, generations=4, children=1000, parents=20, sigma=0.005, tslimit=5000, nact=18, device_id=0, rank=0, world=1,
             comm_from=None, transport=None, deep_chains=(10, 100, 259)):
    """es_distributed's Deep GA (ga.py:136-149, 251-271) through dne_hip.ga.ga_generation: generation 0 evaluates 1000 root
    genomes (normc init), the later ones children of the 20 cached parents."""
    from dne_hip import _lib, ga
    mine = len(ga.shard_children(children, rank, world))
    e = _lib.Engine(_lib.KIND_GA, nact, max_members=max(mine, 1), device_id=device_id, profile_events=True)
    try:
        _share(e, comm_from, world, transport)
        noise.attach(e)
        pop, score, rows = [], np.array([], np.float32), []
        for gen in range(generations):
            e.barrier()
            t0 = time.time()
            pop, score, ln = ga.ga_generation(e, noise.noise.size, sigma, pop, score, children, parents, 1, gen, tslimit, rank, world, transport)
            e.barrier()
            wall = time.time() - t0
            p = e.profile()
            rows.append({"gen": gen, "wall_s": wall, "env_steps": int(ln.sum()), "steps_per_s": float(ln.sum() / wall),
                         "mean_len": float(ln.mean()), "max_len": int(ln.max()), "best": float(score[0]),
                         "rank0_ms": {k: round(p[k], 2) for k in ("eval_ms", "fc_ms", "conv_ms", "env_ms")}})
        # "measure at generations 1, 10, 100 (chain 258 per display.py:31 as stress)".  A run only reaches such
        # chains after that many generations, so: a synthetic population of `parents` genomes carrying 10 / 100 / 259 seeds
        # (RandomState(3)), one generation each on a COLD parent cache (the 20 chains are rebuilt inside the timed call: one streaming
        # pass per chain) and once more warm.  tests/test_gpu_fullsize_configs.py checks the same form against the oracle.
        deep = []
        hi = noise.noise.size - e.P + 1
        for L in deep_chains:
            rs = np.random.RandomState(3 + L)
            popL = [[int(x) for x in rs.randint(0, hi, size=L)] for _ in range(parents)]
            scL = np.zeros(parents, np.float32)
            row = {"chain": L}
            for label in ("cold", "warm"):
                e.barrier()
                t0 = time.time()
                _, _, ln = ga.ga_generation(e, noise.noise.size, sigma, popL, scL, children, parents, 1, 1000 + L, tslimit, rank, world, transport)
                e.barrier()
                wall = time.time() - t0
                row[label] = {"wall_s": wall, "env_steps": int(ln.sum()), "steps_per_s": float(ln.sum() / wall)}
            row["rebuild_ms"] = 1e3 * (row["cold"]["wall_s"] - row["warm"]["wall_s"])
            deep.append(row)
        e.check_redzones()
    finally:
        e.close()
        noise._engines[:] = [x for x in noise._engines if x is not e]
    steady = rows[1:] or rows
    sps = sum(r["env_steps"] for r in steady) / sum(r["wall_s"] for r in steady)
    return {"workload": "FrostbiteNoFrameskip-v4 GA: %d children per generation, top-%d parents, 1 elite, sigma %g, tslimit %d, "
                        "GAAtariPolicy (P=1008450), seed-chain genomes rebuilt on the device" % (children, parents, sigma, tslimit),
            "metric": "env-steps/sec/generation", "value": sps, "unit": "env-steps/s", "n_gpus": world,
            "value_basis": "generations 1..%d (children of cached parents); generation 0 = %d root genomes" % (generations - 1, children),
            "generations": rows, "deep_chains": deep, "roofline": whole_job_roofline(sps, "ga", world)}


def ga_large(noise, generations=3, children=1000, parents=20, power=0.002, tslimit=5000, nact=18, device_id=0):
    """The GPU tree's protocol (gpu_implementation/ga.py:128-176) on its LargeModel: genomes ((idx0,), (idx1, power1), ...),
    scaled-noise root (models/base.py:118-149), truncation to the top `parents` by (-fitness, arrival)."""
    from dne_hip import _lib, ga_gpu
    e = _lib.Engine(_lib.KIND_GA_LARGE, nact, max_members=children, device_id=device_id, profile_events=True)
    try:
        noise.attach(e)
        model = ga_gpu.HipModel(e)
        rs = np.random.RandomState(0)
        cached, rows = [], []
        for gen in range(generations):
            tasks = [model.randomize(rs, noise) if not cached else model.mutate(cached[rs.randint(len(cached))], rs, noise, power)
                     for _ in range(children)]
            seeds = rs.randint(0, 2 ** 32, size=children, dtype=np.uint64).astype(np.uint32)
            t0 = time.time()
            ret, _, ln = e.ga_eval_powers(tasks, tslimit, seeds)
            order = e.ga_select(ret, parents)
            wall = time.time() - t0
            cached = [tasks[i] for i in order]
            p = e.profile()
            rows.append({"gen": gen, "wall_s": wall, "env_steps": int(ln.sum()), "steps_per_s": float(ln.sum() / wall),
                         "mean_len": float(ln.mean()), "max_len": int(ln.max()), "best": float(ret[order[0]]),
                         "ms": {k: round(p[k], 2) for k in ("eval_ms", "fc_ms", "conv_ms", "env_ms")}})
        e.check_redzones()
    finally:
        e.close()
        noise._engines[:] = [x for x in noise._engines if x is not e]
    steady = rows[1:] or rows
    sps = sum(r["env_steps"] for r in steady) / sum(r["wall_s"] for r in steady)
    return {"workload": "FrostbiteNoFrameskip-v4 GA, GPU-tree protocol: LargeModel (conv 32/64/64, fc 512; P=4052658), %d children, "
                        "top-%d parents, mutation power %g, tslimit %d" % (children, parents, power, tslimit),
            "metric": "env-steps/sec/generation", "value": sps, "unit": "env-steps/s", "n_gpus": 1,
            "value_basis": "generations 1..%d" % (generations - 1), "generations": rows,
            "roofline": whole_job_roofline(sps, "ga_large", 1)}


# ------------------------------------------------------------------------------------------------ config 4
def nses(noise, iterations=2, pop=5000, meta_pop=3, archive_extra=29, k=10, tslimit=5000, nact=18, device_id=0, rank=0, world=1,
         comm_from=None, transport=None, algo_type="ns"):
    """NS-ES (nses.py:58-316) co-located: a meta-population of `meta_pop` parameter vectors with their own Adam state, the
    archive seeded with their behaviour characterisations (+ `archive_extra` more so that the novelty pass sees the archive size
    of a run that is `archive_extra` iterations old); per iteration: evaluate the current parent's population with RAM
    trajectories kept in HBM, novelty of all 2N rollouts on the device, all-gather of the 32-byte records, rank blend + update
    on every rank, the parent's new characterisation appended, next parent drawn by novelty (nses.py:293-302)."""
    from dne_hip import _lib, es, nses as N, policies
    cfg = es.Config(**dict(ES_CONFIG, episodes_per_batch=pop, return_proc_mode="centered_sign_rank"))   # configurations/frostbite_nses.json:10
    n_pairs = pop // 2
    e = _es_engine(noise, nact, n_pairs, rank, world, device_id, 0, 0, 0, comm_from, transport, record_bc=True, bc_max_steps=tslimit)
    ref_batch = e._bench_ref
    first_theta = None
    try:
        rs = np.random.RandomState(7)
        thetas, opt_state, archive = {}, {}, []
        t_setup = time.time()
        for p in range(meta_pop + archive_extra):       # nses.py:95-117; the extra entries come from further initialisations
            th = policies.xavier_flat(nact, seed=100 + p)
            e.set_theta(th)
            archive.append(N.get_mean_bc(e, tslimit, rs.randint(2 ** 31)))
            if p == 0:
                first_theta = th
            if p < meta_pop:
                thetas[p] = th
                opt_state[p] = (np.zeros(e.P, np.float32), np.zeros(e.P, np.float32), 0)
        t_setup = time.time() - t_setup
        cur, rows = 0, []
        for it in range(iterations):
            e.barrier()
            t0 = time.time()
            e.set_theta(thetas[cur]); e.optimizer_set_state(*opt_state[cur])
            rec, ratio = N.nses_generation(e, noise.noise.size, cfg, algo_type, archive, k, n_pairs, it, tslimit, ES_OPT, rank, world, transport)
            t_gen = time.time() - t0
            thetas[cur] = e.get_theta(); opt_state[cur] = e.optimizer_get_state()
            archive.append(N.get_mean_bc(e, tslimit, rs.randint(2 ** 31)))                       # nses.py:246-247
            probs = []
            for p in range(meta_pop):                                                            # nses.py:293-302
                e.set_theta(thetas[p])
                probs.append(N.compute_novelty_vs_archive(e, archive, N.get_mean_bc(e, tslimit, rs.randint(2 ** 31)), k))
            probs = np.array(probs) / float(np.sum(probs))
            nxt = int(rs.choice(range(meta_pop), 1, p=probs)[0])
            e.barrier()
            wall = time.time() - t0
            steps = int(rec["len"].sum())
            rows.append({"iteration": it, "parent": cur, "wall_s": wall, "generation_s": t_gen, "env_steps": steps,
                         "steps_per_s": steps / wall, "archive": len(archive), "novelty_mean": float(rec["aux"].mean()),
                         "update_ratio": float(ratio)})
            cur = nxt
        e.check_redzones()
    finally:
        e.close()
        noise._engines[:] = [x for x in noise._engines if x is not e]
    sps = sum(r["env_steps"] for r in rows) / sum(r["wall_s"] for r in rows)
    return {"workload": "Frostbite NS-ES pop=%d, meta-population %d, k=%d, behaviour characterisation = RAM trajectory u8[T,128], "
                        "archive %d -> %d entries resident on the device, tslimit %d" % (pop, meta_pop, k, meta_pop + archive_extra,
                                                                                         meta_pop + archive_extra + iterations, tslimit),
            "metric": "env-steps/sec/iteration (rollouts + novelty + exchange + blend + update + parent selection)", "value": sps,
            "unit": "env-steps/s", "n_gpus": world, "iterations": rows, "archive_setup_s": t_setup,
            "roofline": whole_job_roofline(sps, "es", world),
            "_cpu_inputs": {"theta": first_theta, "ref": ref_batch, "archive": [a.copy() for a in archive[:meta_pop + archive_extra]], "k": k}}


# ------------------------------------------------------------------------------------------------ config 5
def six_games(noise, generations=1, warmup=1, pop=5000, tslimit=5000, games=None, device_id=0, rank=0, world=1, comm_from=None,
              transport=None):
    """Six independent runs of the headline workload, one per game (own theta, reference batch, environment seeds and -- for
    Asteroids -- the 14-action layout); `warmup` untimed + `generations` timed generations each."""
    from dne_hip import es
    cfg = es.Config(**dict(ES_CONFIG, episodes_per_batch=pop))
    n_pairs = pop // 2
    tr = None if transport is None else es.allgather_records
    rows = []
    for gi, game in enumerate(games or list(GAMES)):
        nact = GAMES[game]
        e = _es_engine(noise, nact, n_pairs, rank, world, device_id, gi, 1000 * gi, gi, comm_from, transport)
        try:
            g = 0
            for _ in range(warmup):
                es.es_generation(e, noise.noise.size, cfg, n_pairs, 100 * gi + g, tslimit, ES_OPT, rank, world, tr); g += 1
            e.barrier()
            t0 = time.time(); steps = 0
            for _ in range(generations):
                rec, _ = es.es_generation(e, noise.noise.size, cfg, n_pairs, 100 * gi + g, tslimit, ES_OPT, rank, world, tr); g += 1
                steps += int(rec["len"].sum())
            e.barrier()
            wall = time.time() - t0
            e.check_redzones()
            rows.append({"game": game, "n_actions": nact, "num_params": e.P, "generations": generations, "env_steps": steps,
                         "steps_per_s": steps / wall, "ms_per_generation": 1e3 * wall / generations,
                         "mean_return": float(rec["ret"].mean())})
        finally:
            e.close()
            noise._engines[:] = [x for x in noise._engines if x is not e]
    sps = sum(r["env_steps"] for r in rows) / sum(r["ms_per_generation"] * r["generations"] * 1e-3 for r in rows)
    return {"workload": "6-game Atari ES sweep (%s) pop=%d, tslimit %d, the games one after the other; SynthAtari fixture with each "
                        "game's action count" % ("/".join(r["game"] for r in rows), pop, tslimit),
            "metric": "env-steps/sec/generation", "value": sps, "unit": "env-steps/s", "n_gpus": world, "games": rows,
            "roofline": whole_job_roofline(sps, "es", world)}


# ------------------------------------------------------------------------------------------------ N ranks rehearsed on one GPU
def simulate_ranks(noise, world, steps=20, warmup=5, pop=5000, tslimit=5000, nact=18, device_id=0, verify_generations=2, exp=None, shard=None):
    """An N-GPU run of the headline workload rehearsed on ONE GPU: per generation the N round-robin shards
    (es.shard_pairs, the index stream of rank r = RandomState(generation * N + r) exactly as an N-rank launch draws it) are evaluated one
    after the other on one engine sized for a rank's share, each through dne_records_pack; the records are put back into global pair order
    exactly as the exchange's receive side does (es.allgather_records / dne_allgather_results) and every "rank" would then run the same
    redundant update (dne_es_update_gathered, once here).  Per generation the N-GPU wall time is predicted as the SLOWEST shard's
    evaluation + the update (the all-gather of N x 32-byte records over xGMI is tens of microseconds and is not on this box to measure):
        value = sum of all shards' env-steps / sum over timed generations of (max over ranks of eval seconds + update seconds).
    What the rehearsal PROVES is the data path: for the first `verify_generations` a second engine evaluates the union of the shards' pairs
    as ONE rank (same indices and seeds, in global pair order) and the theta digests must agree -- sharding, packing, un-sharding and the
    gathered update change nothing.  The one thing an N-GPU box adds is ncclCommInitRank + the collective itself.
    Reference: gpu_implementation/neuroevolution/concurrent_worker.py:128-142 (one worker per device), es_distributed/es.py:377-439."""
    import hashlib
    from dne_hip import es
    n_pairs = pop // 2
    exp = exp or {"config": {"calc_obstat_prob": 0.0, "episodes_per_batch": pop, "eval_prob": 0.0, "l2coeff": 0.005, "noise_stdev": 0.02,
                             "snapshot_freq": 0, "timesteps_per_batch": 10000, "return_proc_mode": "centered_rank", "episode_cutoff_mode": 5000},
                  "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"}}
    config = es.Config(**exp["config"])
    opt = exp["optimizer"]
    share = len(es.shard_pairs(n_pairs, 0, world))
    e = _es_engine(noise, nact, share, 0, 1, device_id, 0, 0, 0, None, None)           # a rank's engine: max_members = 2 x its share
    v = _es_engine(noise, nact, n_pairs, 0, 1, device_id, 0, 0, 0, None, None) if verify_generations > 0 else None
    P = e.P
    sha = lambda eng: hashlib.sha256(eng.get_theta().tobytes()).hexdigest()
    agree = []
    per_gen = []
    rank_ms = np.zeros(world)
    steps_total = 0
    t_pred = 0.0
    for g in range(warmup + steps):
        full = np.zeros(n_pairs, es.RECORD)
        idx_all = np.zeros(n_pairs, np.int64); seeds_all = np.zeros(2 * n_pairs, np.uint32)
        evals, gsteps = [], 0
        for r in range(world):
            mine, idx, seeds = es.generation_inputs(noise.noise.size, P, n_pairs, g, r, world, shard=shard)
            t0 = time.time()
            e.es_eval(idx, config.noise_stdev, tslimit, seeds)
            rec = e.records_pack(len(mine))
            evals.append(time.time() - t0)
            full[mine] = rec[:len(mine)]
            idx_all[mine] = idx; seeds_all[2 * mine] = seeds[0::2]; seeds_all[2 * mine + 1] = seeds[1::2]
            gsteps += int(e.profile()["env_steps"])
        t0 = time.time()
        e.records_set(full)
        e.es_update_gathered(config.return_proc_mode, opt["type"], config.l2coeff, *es.optimizer_args(opt))
        e.barrier()          # hipDeviceSynchronize (no communicator here)
        upd = time.time() - t0
        if v is not None and g < verify_generations:
            v.es_eval(idx_all, config.noise_stdev, tslimit, seeds_all)
            v.records_set(v.records_pack(n_pairs))
            v.es_update_gathered(config.return_proc_mode, opt["type"], config.l2coeff, *es.optimizer_args(opt))
            agree.append(sha(e) == sha(v))
        if g >= warmup:
            steps_total += gsteps
            t_pred += max(evals) + upd
            rank_ms += 1e3 * np.asarray(evals)
            per_gen.append(round(1e3 * (max(evals) + upd), 2))
    digest = sha(e)
    e.close()
    if v is not None:
        v.close()
    if agree and not all(agree):
        raise RuntimeError("simulate_ranks(%d): theta after the sharded generations differs from the one-rank evaluation of the same pairs: %s" % (world, agree))
    return {"workload": "the headline workload as %d round-robin shards of %d pairs evaluated one after the other on one GPU" % (world, share),
            "metric": "env-steps/sec/generation (predicted for %d GPUs)" % world, "value": steps_total / t_pred, "unit": "env-steps/s",
            "n_gpus_simulated": world, "pairs_per_rank": share, "shard": shard or es.shard_mode(), "steps": steps, "warmup": warmup,
            "ms_per_step": 1e3 * t_pred / steps, "rank_eval_ms_mean": [round(x / steps, 2) for x in rank_ms],
            "rank0_share_ms_per_generation": round(rank_ms[0] / steps, 2), "update_included": True, "exchange_included": False,
            "theta_matches_one_rank_evaluation": (all(agree) if agree else None), "verified_generations": len(agree), "theta_sha256": digest,
            "basis": "max over the simulated ranks of (evaluation + dne_records_pack) + one gathered update, per generation; the RCCL all-gather "
                     "itself (N x 32-byte records) is not executed"}


# ------------------------------------------------------------------------------------------------ CPU legs (oracle = checker)
_BASE = None


def _cpu_es_pair(i):
    os.environ["OMP_NUM_THREADS"] = "1"
    import oracle as O
    noise, theta, ref, sigma, tslimit, nact, idx, seeds = _BASE
    L = O.layout(O.KIND_ES, nact)
    t0, c0 = time.time(), time.process_time()
    _, _, ln = O.es_eval(L, theta, noise, idx[i:i + 1], sigma, tslimit, ref, seeds[2 * i:2 * i + 2])
    return int(ln.sum()), time.time() - t0, time.process_time() - c0


def _cpu_ga_child(i):
    os.environ["OMP_NUM_THREADS"] = "1"
    import oracle as O
    noise, sigma, tslimit, nact, fresh, seeds = _BASE
    L = O.layout(O.KIND_GA, nact)
    t0, c0 = time.time(), time.process_time()
    r = O.rollout(L, O.ga_rebuild(L, noise, [int(fresh[i])], sigma), None, seeds[i], tslimit)
    return int(r[2]), time.time() - t0, time.process_time() - c0


def _cpu_pool_warm(_):
    os.environ["OMP_NUM_THREADS"] = "1"
    import oracle as O
    O.lib()
    time.sleep(0.02)    # long enough that the start-up tasks spread over all workers
    return os.getpid()


def _pool_rate(fn, n, procs):
    """n work items over `procs` forked single-threaded workers -> (env-steps, CPU seconds the workers consumed
    [time.process_time inside each item: what the cores really delivered], wall seconds of the items alone: the pool is started and
    every worker has loaded the oracle BEFORE the clock starts -- the reference's workers are long-lived processes (es.py:366), their
    start-up is not part of a generation; stragglers at the end are)"""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.lib()
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_cpu_pool_warm, range(4 * procs), chunksize=1)
        t0 = time.time()
        res = pool.map(fn, range(n), chunksize=1)
        wall = time.time() - t0
    return int(sum(r[0] for r in res)), float(sum(r[2] for r in res)), wall


def _sweep_counts(procs):
    """worker counts to try: the reference runs one worker per core it is given (launch.py:117) -- what the cgroup / affinity mask
    admits -- and twice that (whether oversubscribing the granted CPUs pays is a property of the host)"""
    from hostinfo import usable_cpus
    if procs:
        return [int(procs)]
    u = max(1, min(usable_cpus(), os.cpu_count() or 1))
    return [u, 2 * u]


def _cpu_sweep(fn, counts, n_total, items_for):
    """time `fn` at every worker count; the baseline is the BEST wall-clock rate (the reference's operator would pick that count)"""
    rows = []
    for w in counts:
        n = min(items_for(w), n_total)
        steps, cpu_s, wall = _pool_rate(fn, n, w)
        rows.append({"workers": w, "items": n, "env_steps": steps, "wall_s": round(wall, 2), "cpu_s": round(cpu_s, 2),
                     "rate_wall": steps / wall, "rate_per_cpu_second": steps / max(cpu_s, 1e-9),
                     "cpus_delivered": round(cpu_s / wall, 1)})
    return rows, max(rows, key=lambda r: r["rate_wall"])


def cpu_es(noise, theta, ref, sigma, tslimit, nact, n_pairs_total=2500, procs=None, sample_pairs=None, generation=0):
    """The CPU oracle structured like the reference workers (one single-threaded process per worker, one antithetic pair at a
    time, batch-1 forwards, a reference pass per episode: es.py:366-439, launch.py:117) on the first pairs of a generation, at a
    sweep of worker counts.  `value` is the best WALL-CLOCK rate of the sweep (process start-up and stragglers included) and
    `cores` the worker count that gave it; `host` says what the box really offers (affinity mask, cgroup quota, CPU model),
    `cpus_delivered` = CPU seconds consumed / wall seconds of that leg."""
    global _BASE
    from dne_hip import es
    from hostinfo import host_facts
    _, idx, seeds = es.generation_inputs(noise.size, theta.size, n_pairs_total, generation, 0, 1)
    _BASE = (noise, theta, ref, sigma, tslimit, nact, idx, seeds)
    # >= 20 pairs per worker (about 5 s of wall at ~1.2 k env-steps per CPU-second and ~270 env-steps per pair), never fewer than 256
    # pairs: round 5's 32-pair, 0.6-second sample moved by 24 % from box to box
    items = (lambda w: sample_pairs) if sample_pairs else (lambda w: min(max(20 * w, 256), 640))
    rows, best = _cpu_sweep(_cpu_es_pair, _sweep_counts(procs), n_pairs_total, items)
    return {"value": best["rate_wall"], "unit": "env-steps/s", "cores": best["workers"], "kind": "port",
            "cpus_delivered": best["cpus_delivered"], "rate_per_cpu_second": best["rate_per_cpu_second"],
            "sample": "first %d antithetic pairs of generation %d (%d full episodes, %d env-steps) over %d single-threaded worker "
                      "processes started (oracle loaded) before the clock: %.1f s wall, %.1f CPU-seconds; value = env-steps / wall of the best worker count of the sweep"
                      % (best["items"], generation, 2 * best["items"], best["env_steps"], best["workers"], best["wall_s"], best["cpu_s"]),
            "sweep": rows, "host": host_facts()}


def _cpu_nses_pair(i):
    os.environ["OMP_NUM_THREADS"] = "1"
    import oracle as O
    noise, theta, ref, sigma, tslimit, nact, idx, seeds, archive, k = _BASE
    L = O.layout(O.KIND_ES, nact)
    t0, c0 = time.time(), time.process_time()
    steps = 0
    for s in range(2):   # nses.py:371-384: both rollouts keep their RAM trajectory and are scored against the archive
        th = O.perturb(theta, noise, idx[i], sigma, 1 if s == 0 else -1)
        _, _, ln, bc = O.rollout(L, th, ref, seeds[2 * i + s], tslimit, want_bc=True)
        O.novelty(archive, bc, k)
        steps += ln
    return int(steps), time.time() - t0, time.process_time() - c0


def cpu_nses(noise, theta, ref, archive, k, sigma, tslimit, nact, n_pairs_total=2500, procs=None, sample_pairs=None):
    """nses.py:318-400's worker on the oracle: an antithetic pair with RAM trajectories + the novelty of both rollouts against the
    archive; one worker count (the headline sweep's best) unless told otherwise"""
    global _BASE
    from dne_hip import es
    _, idx, seeds = es.generation_inputs(noise.size, theta.size, n_pairs_total, 0, 0, 1)
    _BASE = (noise, theta, ref, sigma, tslimit, nact, idx, seeds, archive, k)
    items = (lambda w: sample_pairs) if sample_pairs else (lambda w: min(max(2 * w, 16), 128))
    rows, best = _cpu_sweep(_cpu_nses_pair, _sweep_counts(procs), n_pairs_total, items)
    return {"value": best["rate_wall"], "unit": "env-steps/s", "cores": best["workers"], "kind": "port", "cpus_delivered": best["cpus_delivered"],
            "sample": "first %d antithetic pairs of iteration 0 with trajectories + novelty (k = %d, archive of %d): %d env-steps over %d worker "
                      "processes, %.1f s wall, %.1f CPU-seconds" % (best["items"], k, len(archive), best["env_steps"], best["workers"],
                                                                      best["wall_s"], best["cpu_s"]), "sweep": rows}


def cpu_ga(noise, sigma, tslimit, nact, children=1000, procs=None, sample=None):
    """ga.py:209-271's worker on the oracle: rebuild a root genome (normc), one episode; the first children of generation 0,
    at the same sweep of worker counts as cpu_es"""
    global _BASE
    from dne_hip import _lib, ga
    _, _, fresh, seeds = ga.ga_generation_inputs(noise.size, _lib.num_params(_lib.KIND_GA, nact), children, 0, 0, 0, 1)
    _BASE = (noise, sigma, tslimit, nact, fresh, seeds)
    items = (lambda w: sample) if sample else (lambda w: min(max(w, 16), 256))
    rows, best = _cpu_sweep(_cpu_ga_child, _sweep_counts(procs), children, items)
    return {"value": best["rate_wall"], "unit": "env-steps/s", "cores": best["workers"], "kind": "port",
            "cpus_delivered": best["cpus_delivered"],
            "sample": "first %d children of generation 0 (%d env-steps) over %d single-threaded worker processes: %.1f s wall, "
                      "%.1f CPU-seconds; best worker count of the sweep" % (best["items"], best["env_steps"], best["workers"],
                                                                            best["wall_s"], best["cpu_s"]),
            "sweep": rows}


def cpu_sweep(noise, games, cpu_ref, tslimit, procs=None, sample_pairs=None):
    """config 5 on the CPU: the fixture stands in for every emulator, so the six games are two workloads -- the 18-action network
    (= the headline's CPU baseline, reused) and Asteroids' 14-action one (timed here) -- combined over the games that ran as
    total env-steps / total time (equal steps per game)."""
    from dne_hip import policies
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    rates, rows = {}, {}
    if cpu_ref:
        rates[18] = cpu_ref["value"]
    for nact in sorted({g["n_actions"] for g in games}):
        if nact in rates:
            continue
        gi = [g["game"] for g in games if g["n_actions"] == nact][0]
        r = cpu_es(noise, policies.xavier_flat(nact, seed=2), O.get_ref_batch(seed=2, batch_size=128, nact=nact), 0.02, tslimit, nact,
                   procs=procs, sample_pairs=sample_pairs)
        rates[nact], rows[gi] = r["value"], {k: r[k] for k in ("value", "cores", "cpus_delivered", "sample")}
    inv = sum(1.0 / rates[g["n_actions"]] for g in games)
    return {"value": len(games) / inv, "unit": "env-steps/s", "kind": "port", "cores": (cpu_ref or {}).get("cores") or procs,
            "per_action_count": {str(k): v for k, v in rates.items()}, "timed_here": rows,
            "sample": "18-action games: the headline's cpu_baseline; other action counts timed here; combined over %d games" % len(games)}


def config1_cpu(noise, nact=18, pop=256, sigma=0.02, tslimit=5000, sample_pairs=8):
    """BASELINE config 1: the reference CPU path -- frostbite_es.json schema with pop 256 and sigma 0.02, exactly 2 worker
    processes (scripts/local_run_exp.sh:10) -- on the oracle, timed on a bounded sample and extrapolated to the generation."""
    from dne_hip import policies
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    theta = policies.xavier_flat(nact, seed=0)
    ref = O.get_ref_batch(seed=0, batch_size=128, nact=nact)
    r = cpu_es(noise, theta, ref, sigma, tslimit, nact, n_pairs_total=pop // 2, procs=2, sample_pairs=sample_pairs)
    r["workload"] = "frostbite_es.json schema, pop=%d (N=%d pairs), sigma=%g, 2 CPU worker processes, no GPU" % (pop, pop // 2, sigma)
    return r
