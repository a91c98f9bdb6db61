"""CPU, world_size 2 over gloo: the N > 1 path (shard pairs round-robin -> evaluate -> all-gather 32-byte
records -> redundant update) leaves bit-identical theta on every rank, equal to a serial emulation."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PAIRS, TSLIMIT, GENS, NOISE = 6, 10, 2, 2_500_000
OPT = {"type": "adam", "args": {"stepsize": 0.01}}


def _config():
    from dne_hip import es
    return es.Config(l2coeff=0.005, noise_stdev=0.02, episodes_per_batch=2 * N_PAIRS, timesteps_per_batch=10,
                     calc_obstat_prob=0.0, eval_prob=0.0, snapshot_freq=0, return_proc_mode="centered_rank",
                     episode_cutoff_mode=TSLIMIT)


def _make_engine():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from oracle_engine import OracleEngine
    from dne_hip import policies
    eng = OracleEngine(0, ref_count=16)
    eng.noise_upload(np.random.RandomState(123).randn(NOISE).astype(np.float32))
    eng.set_theta(policies.xavier_flat(18, 0))
    eng.set_ref_batch(O.get_ref_batch(seed=0, batch_size=16))
    return eng


def _rank_main(rank, world, port, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "deep-neuroevolution_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from dne_hip import es
    eng = _make_engine()
    recs = []
    for g in range(GENS):
        rec, ratio = es.es_generation(eng, NOISE, _config(), N_PAIRS, g, TSLIMIT, OPT, rank, world, transport=es.allgather_records)
        recs.append(rec.tobytes())
    q.put((rank, eng.get_theta().tobytes(), recs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_gloo_bit_identical():
    import torch.multiprocessing as mp
    from dne_hip import es
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=500) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, th0, rec0), (_, th1, rec1) = out
    assert th0 == th1 and rec0 == rec1                      # every rank holds the same theta and the same records
    # serial emulation: one engine evaluates both shards, merges in global pair order, updates once
    eng = _make_engine()
    cfg = _config()
    for g in range(GENS):
        full = np.zeros(N_PAIRS, es.RECORD)
        for r in range(2):
            mine, idx, seeds = es.generation_inputs(NOISE, eng.P, N_PAIRS, g, r, 2)
            ret, sg, ln = eng.es_eval(idx, cfg.noise_stdev, TSLIMIT, seeds)
            full[mine] = es.pack_records(idx, ret, ln, sg)
        assert full.tobytes() == rec0[g]
        eng.es_update(full["noise_idx"], full["ret"], full["aux"], "centered_rank", "adam", cfg.l2coeff, 0.01)
    assert eng.get_theta().tobytes() == th0
    rec = np.frombuffer(rec0[-1], es.RECORD)
    assert rec["len"].min() >= 1 and rec["len"].max() <= TSLIMIT and rec["ret"].dtype == np.float32


# ------------------------------------------------------------------------------------------------ Deep GA, two ranks
GA_CHILDREN, GA_PARENTS, GA_GENS = 7, 3, 3


def _ga_engine():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_engine import OracleEngine
    eng = OracleEngine(1)
    eng.noise_upload(np.random.RandomState(123).randn(NOISE).astype(np.float32))
    return eng


def _gloo_allgather_bytes(buf, world):
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(buf.view(np.uint8).copy())
    out = torch.empty(world * t.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, t)
    return out.numpy().view(buf.dtype).reshape(world, -1)


def _ga_rank_main(rank, world, port, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "deep-neuroevolution_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from dne_hip import ga
    eng = _ga_engine()
    pop, score = [], np.array([], np.float32)
    for g in range(GA_GENS):
        pop, score, _ = ga.ga_generation(eng, NOISE, 0.005, pop, score, GA_CHILDREN, GA_PARENTS, 1, g, 12, rank, world,
                                         transport=_gloo_allgather_bytes)
    q.put((rank, repr(pop), score.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ga_two_ranks_gloo_same_elites():
    """Deep GA with the children sharded over two ranks (odd count: 4 + 3): after the all-gather of 32-byte child records
    every rank holds the same population, equal to a one-rank emulation of both shards (ga.py:136-149, 251-271)."""
    import torch.multiprocessing as mp
    from dne_hip import ga
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ga_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=500) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1:] == out[1][1:]
    eng = _ga_engine()
    pop, score = [], np.array([], np.float32)
    for g in range(GA_GENS):      # serial emulation: evaluate both shards on one engine, merge in global child order
        recs = np.zeros(GA_CHILDREN, ga.CHILD_RECORD)
        for r in range(2):
            mine, parent, fresh, env_seeds = ga.ga_generation_inputs(NOISE, eng.P, GA_CHILDREN, len(pop), g, r, 2)
            chains = [(list(pop[p]) if p >= 0 else []) + [int(f)] for p, f in zip(parent, fresh)]
            ret, sg, ln = eng.ga_eval(chains, 0.005, 12, env_seeds)
            recs["parent"][mine], recs["seed"][mine], recs["ret"][mine], recs["len"][mine] = parent, fresh, ret, ln
        cand = [list(c) for c in pop[:1]] + [(list(pop[r["parent"]]) if r["parent"] >= 0 else []) + [int(r["seed"])] for r in recs]
        cand_ret = np.array(list(score[:1]) + list(recs["ret"]), np.float32)
        pop, score = ga.truncate(eng, cand, cand_ret, GA_PARENTS)
    assert repr(pop) == out[0][1] and score.tobytes() == out[0][2]
    assert len(pop) == GA_PARENTS and all(len(c) >= 1 for c in pop) and max(len(c) for c in pop) >= 2


# ------------------------------------------------------------------------------------------------ NS-ES / NSR-ES, two ranks
def _nses_engine():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from oracle_engine import OracleEngine
    from dne_hip import policies
    eng = OracleEngine(0, ref_count=16, bc_max_steps=TSLIMIT)
    eng.noise_upload(np.random.RandomState(123).randn(NOISE).astype(np.float32))
    eng.set_theta(policies.xavier_flat(18, 0))
    eng.set_ref_batch(O.get_ref_batch(seed=0, batch_size=16))
    return eng


def _archive():
    rs = np.random.RandomState(77)
    return [rs.randint(0, 256, (n, 128)).astype(np.uint8) for n in (9, 4, 10)]


def _nses_config():
    from dne_hip import es
    return es.Config(l2coeff=0.005, noise_stdev=0.02, episodes_per_batch=2 * N_PAIRS, timesteps_per_batch=10, calc_obstat_prob=0.0,
                     eval_prob=0.0, snapshot_freq=0, return_proc_mode="centered_sign_rank", episode_cutoff_mode=TSLIMIT)


def _nses_rank_main(rank, world, port, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "deep-neuroevolution_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from dne_hip import nses
    eng = _nses_engine()
    out = []
    for g, algo in enumerate(("ns", "nsr")):
        rec, _ = nses.nses_generation(eng, NOISE, _nses_config(), algo, _archive(), 2, N_PAIRS, g, TSLIMIT, OPT, rank, world,
                                      transport=_gloo_allgather_bytes)
        out.append(rec.tobytes())
    q.put((rank, eng.get_theta().tobytes(), out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_nses_two_ranks_gloo_bit_identical():
    """Config 4's exchange: novelty travels in the records' aux slot (nses.py:384), the rank blend (nses.py:217-228) and the
    update run redundantly -- same theta on both ranks, equal to a one-rank run of the same two generations (NS then NSR)."""
    import torch.multiprocessing as mp
    from dne_hip import es, nses
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nses_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=500) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1:] == out[1][1:]
    eng = _nses_engine()
    cfg = _nses_config()
    for g, algo in enumerate(("ns", "nsr")):          # serial emulation of both shards
        full = np.zeros(N_PAIRS, es.RECORD)
        for r in range(2):
            mine, idx, seeds = es.generation_inputs(NOISE, eng.P, N_PAIRS, g, r, 2)
            ret, _, ln = eng.es_eval(idx, cfg.noise_stdev, TSLIMIT, seeds)
            nov = eng.novelty_batch(_archive(), ln, 2).astype(np.float32).reshape(-1, 2)
            full[mine] = es.pack_records(idx, ret, ln, nov)
        assert full.tobytes() == out[0][2][g]
        nses.blend_and_update(eng, full, algo, cfg.return_proc_mode, cfg.l2coeff, OPT)
    assert eng.get_theta().tobytes() == out[0][1]
    rec = np.frombuffer(out[0][2][1], es.RECORD)
    assert (rec["aux"] > 0).all()                      # novelty, not sign-returns


# ------------------------------------------------------------------------------------------------ bench.py's own launcher, 4 and 8 ranks
def _launch_oracle_ranks(tmp_path, world, mode, n_pairs, tsl, nref):
    """bench.launch_ranks (the pipes, RANK / WORLD_SIZE / MASTER_* of the ranks) with tests/rank_stub_oracle.py as the rank"""
    import argparse
    import json
    sys.path.insert(0, ROOT)
    import bench
    os.environ["STUB_OUT"] = str(tmp_path)
    try:
        args = argparse.Namespace(gpus=world, single_device=True, transport="gloo")
        rc = bench.launch_ranks(args, child_argv=[sys.executable, os.path.join(ROOT, "tests", "rank_stub_oracle.py"), mode, str(n_pairs), str(tsl),
                                                  str(nref)], check_devices=False)
    finally:
        del os.environ["STUB_OUT"]
    assert rc == 0
    return [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(world)]


@pytest.mark.timeout(900)
def test_eight_ranks_full_population_shards(tmp_path):
    """SURVEY 8e at the driver's width: 2500 pairs over 8 ranks = shards of 313 / 312 (round-robin), one all-gather of 32-byte
    records (gloo here), the redundant update -- every rank ends with the same records and the same theta; the gathered array is in
    global pair order whatever rank produced an entry (spot-checked against a one-rank evaluation of those pairs)."""
    from dne_hip import es
    n_pairs, tsl, nref = 2500, 3, 4
    out = _launch_oracle_ranks(tmp_path, 8, "es", n_pairs, tsl, nref)
    assert [o["mine"] for o in out] == [313] * 4 + [312] * 4 and [o["evaluated"] for o in out] == [o["mine"] for o in out]
    assert all(o["n_records"] == n_pairs for o in out)
    assert len({o["records"] for o in out}) == 1 and len({o["theta"] for o in out}) == 1
    # pairs 0, 1, 2 belong to ranks 0, 1, 2; the last two to ranks (2498 % 8, 2499 % 8) = (2, 3): re-derive them on one engine
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from oracle_engine import OracleEngine
    from dne_hip import policies
    eng = OracleEngine(0, ref_count=nref)
    eng.noise_upload(np.random.RandomState(123).randn(2_500_000).astype(np.float32))
    eng.set_theta(policies.xavier_flat(18, 0)); eng.set_ref_batch(O.get_ref_batch(seed=0, batch_size=nref))
    want = np.zeros(n_pairs, es.RECORD)
    for r in (0, 1, 2, 3):
        mine, idx, seeds = es.generation_inputs(2_500_000, eng.P, n_pairs, 0, r, 8)
        sel = [k for k, g in enumerate(mine) if g in (0, 1, 2, 2498, 2499)]
        for k in sel:
            ret, sg, ln = eng.es_eval(idx[k:k + 1], 0.02, tsl, seeds[2 * k:2 * k + 2])
            want[mine[k]] = es.pack_records(idx[k:k + 1], ret, ln, sg)[0]
    assert want[:3].tobytes().hex() == out[0]["first"] and want[-2:].tobytes().hex() == out[0]["last"]


@pytest.mark.timeout(900)
def test_four_ranks_nses_through_the_launcher(tmp_path):
    """BASELINE config 4 is defined on 4 GPUs: NSR-ES with the population sharded 4 ways (novelty in the records' aux slot,
    nses.py:384; blend + update on every rank, nses.py:217-228) through bench.py's launcher -- identical records and theta on the
    four ranks, equal to a one-rank run of the same generation."""
    from dne_hip import es, nses
    n_pairs, tsl, nref = 22, 8, 8                     # shards of 6, 6, 5, 5
    out = _launch_oracle_ranks(tmp_path, 4, "nses", n_pairs, tsl, nref)
    assert [o["mine"] for o in out] == [6, 6, 5, 5] and len({o["records"] for o in out}) == 1 and len({o["theta"] for o in out}) == 1
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import oracle as O
    from oracle_engine import OracleEngine
    from dne_hip import policies
    eng = OracleEngine(0, ref_count=nref, bc_max_steps=tsl)
    eng.noise_upload(np.random.RandomState(123).randn(2_500_000).astype(np.float32))
    eng.set_theta(policies.xavier_flat(18, 0)); eng.set_ref_batch(O.get_ref_batch(seed=0, batch_size=nref))
    rs = np.random.RandomState(77)
    archive = [rs.randint(0, 256, (n, 128)).astype(np.uint8) for n in (9, 4, 10)]
    full = np.zeros(n_pairs, es.RECORD)
    for r in range(4):
        mine, idx, seeds = es.generation_inputs(2_500_000, eng.P, n_pairs, 0, r, 4)
        ret, _, ln = eng.es_eval(idx, 0.02, tsl, seeds)
        full[mine] = es.pack_records(idx, ret, ln, eng.novelty_batch(archive, ln, 2).astype(np.float32).reshape(-1, 2))
    assert hashlib.sha256(full.tobytes()).hexdigest() == out[0]["records"]
    nses.blend_and_update(eng, full, "nsr", "centered_sign_rank", 0.005, OPT)
    assert hashlib.sha256(eng.get_theta().tobytes()).hexdigest() == out[0]["theta"]
