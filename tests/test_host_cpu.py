"""CPU (-m "not gpu"): the C-ABI library loads and exports every declared symbol, the host mirror of the
reference interface behaves like the reference drivers, and the engine refuses to run without a GPU."""
import ctypes
import os
import re
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from dne_hip import _lib
    hdr = open(os.path.join(ROOT, "include", "dne_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dne_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.build())
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # pure-host entry point (no GPU needed): Policy.num_params of both Atari policies
    assert lib.dne_num_params(0, 18) == 1009058 and lib.dne_num_params(1, 18) == 1008450
    assert lib.dne_num_params(0, 14) == 1009058 - 4 * 257


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from dne_hip import _lib
    with pytest.raises(_lib.DneError) as ei:
        _lib.Engine(_lib.KIND_ES, 18, max_members=4)
    assert "no HIP device" in str(ei.value) or "hip" in str(ei.value).lower()


def test_product_does_not_import_the_oracle():
    import ast
    pkg = os.path.join(ROOT, "deep-neuroevolution_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "dne_oracle" not in src and "libdne_oracle" not in src, f
    # bench.py reaches the oracle only through the CPU-baseline legs of tools/workloads.py (cpu_es / cpu_ga / config1_cpu and
    # their pool workers); nothing that runs on the GPU imports it
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "import oracle" not in bench and "def cpu_baseline" in bench
    tree = ast.parse(open(os.path.join(ROOT, "tools", "workloads.py")).read())
    users = set()
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, ast.Import) and any(a.name == "oracle" for a in node.names):
                users.add(fn.name)
    assert users and all("cpu" in name or name == "_pool_rate" for name in users), users
    assert not any(isinstance(n, ast.Import) and any(a.name == "oracle" for a in n.names) for n in tree.body)


def test_flat_layout_and_init_match_oracle(oracle):
    from dne_hip import _lib, policies
    for kind, nact in ((0, 18), (1, 18), (0, 14)):
        spec, P = policies.flat_layout(kind, nact)
        L = oracle.layout(kind, nact)
        assert P == L.P
        names = {0: ("conv1/weights", "conv2/weights", "fc/weights", "out/weights", "out/biases"),
                 1: ("conv1/w", "conv2/w", "fc/w", "out/w", "out/b")}[kind]
        assert [spec[n][0] for n in names] == [L.c1w, L.c2w, L.fcw, L.ow, L.ob]
    assert np.array_equal(policies.xavier_flat(18, 0), oracle.es_init_theta(oracle.layout(0, 18), 0))
    spec, _ = policies.flat_layout(0, 18)     # SURVEY 8 offsets
    assert spec["fc/weights"][0] == 12432 and spec["BatchNorm_2/gamma"][0] == 1004176 and spec["out/biases"][0] == 1009040


def test_noise_table_and_sharding(small_noise, golden):
    from dne_hip import es
    t = es.SharedNoiseTable(count=100_000, seed=123)
    assert np.array_equal(t.noise, small_noise[:100_000])
    assert t.get(5, 7).base is t.noise                       # a view, like es.py:63-64
    rs = np.random.RandomState(0)

    class Big(es.SharedNoiseTable):
        def __init__(self):
            self.noise = np.lib.stride_tricks.as_strided(np.zeros(1, np.float32), (250_000_000,), (0,))
    assert [Big().sample_index(rs, 1009058) for _ in range(4)] == [209652396, 130329135, 118924917, 136432832]
    for world in (1, 2, 3, 8):
        ids = np.concatenate([es.shard_pairs(2500, r, world) for r in range(world)])
        assert np.array_equal(np.sort(ids), np.arange(2500))
    m0, i0, s0 = es.generation_inputs(4_000_000, 1009058, 10, 3, 0, 2)
    m1, i1, s1 = es.generation_inputs(4_000_000, 1009058, 10, 3, 1, 2)
    assert m0.tolist() == [0, 2, 4, 6, 8] and m1.tolist() == [1, 3, 5, 7, 9]
    _, ia, sa = es.generation_inputs(4_000_000, 1009058, 10, 3, 0, 1)
    both = np.zeros(20, np.uint32); both[np.repeat(2 * m0, 2) + np.tile([0, 1], 5)] = s0; both[np.repeat(2 * m1, 2) + np.tile([0, 1], 5)] = s1
    assert np.array_equal(both, sa)                            # env seeds depend on the global pair id only
    assert es.RECORD.itemsize == 32
    # world > 1: 'table' (default) = rank r draws inside the r-th of `world` stretches of the legal starts, 'uniform' = the whole table as a
    # reference worker does (es.py:412); one rank: the same draw either way, and the legacy stream's values
    hi = 4_000_000 - 1009058 + 1
    assert es.shard_mode() == "table" and i0.max() < hi // 2 <= i1.min() and i1.max() < hi
    for w in (2, 3, 8):
        edges = [es.index_range(4_000_000, 1009058, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == hi and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    _, iu, su = es.generation_inputs(4_000_000, 1009058, 10, 3, 1, 2, shard="uniform")
    assert iu.tolist() == sorted(np.random.RandomState(3 * 2 + 1).randint(0, hi, size=5).tolist()) and np.array_equal(su, s1)
    assert np.array_equal(es.generation_inputs(4_000_000, 1009058, 10, 3, 0, 1, shard="uniform")[1], ia)
    assert es.index_range(4_000_000, 1009058, 0, 1, "table") == (0, hi) == es.index_range(4_000_000, 1009058, 1, 2, "uniform")
    # the index draw of generation_inputs = successive sample_index calls on the same stream (es.py:412), in ascending order
    rs2 = np.random.RandomState(3 * 1 + 0)
    t4 = es.SharedNoiseTable(count=100_000, seed=123)
    _, iv, _ = es.generation_inputs(t4.noise.size, 1000, 64, 3, 0, 1)
    assert iv.tolist() == sorted(t4.sample_index(rs2, 1000) for _ in range(64))   # the worker's draws, labelled in table order


def _exp(pop, tslimit):
    return {"config": {"calc_obstat_prob": 0.0, "episodes_per_batch": pop, "eval_prob": 0.0, "l2coeff": 0.005,
                       "noise_stdev": 0.02, "snapshot_freq": 0, "timesteps_per_batch": 10,
                       "return_proc_mode": "centered_rank", "episode_cutoff_mode": tslimit},
            "env_id": "FrostbiteNoFrameskip-v4", "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"},
            "policy": {"args": {}, "type": "ESAtariPolicy"}}


def test_run_master_run_worker_in_process(oracle, small_noise, tmp_path):
    """The reference's master/worker protocol (declare task -> Results -> update), with the oracle standing
    in for the GPU engine: two iterations must reproduce a hand-rolled oracle ES loop bit-for-bit."""
    from oracle_engine import OracleEngine
    from dne_hip import dist, es
    dist.reset_brokers()
    exp = _exp(pop=8, tslimit=12)
    noise = es.SharedNoiseTable(count=2_500_000)
    master_engine, worker_engine = OracleEngine(0, ref_count=16), OracleEngine(0, ref_count=16)
    cfg = {"unix_socket_path": "/tmp/test_master.sock", "transport": "inprocess"}
    out = {}

    def master():
        out["policy"] = es.run_master(cfg, str(tmp_path), exp, engine=master_engine, noise=noise, max_iters=2, seed=0)

    tm = threading.Thread(target=master, daemon=True)
    tm.start()
    es.run_worker(cfg, cfg, noise, engine=worker_engine, max_tasks=2, seed=7, reeval_after=1e9)
    tm.join(timeout=120)
    assert not tm.is_alive()
    theta = out["policy"].get_trainable_flat()
    # replay with the oracle directly
    from dne_hip import policies
    L = oracle.layout(0, 18)
    th = policies.xavier_flat(18, 0)
    ref = master_engine.ref
    rs = np.random.RandomState(7); rs.randint(2 ** 31)
    opt = oracle.Adam(th, 0.01)
    for it in range(2):
        rs.rand()
        idx = np.sort(np.array([noise.sample_index(rs, L.P) for _ in range(4)], np.int64))   # the worker labels its draws in table order
        seeds = rs.randint(0, 2 ** 32, size=8, dtype=np.uint64).astype(np.uint32)
        rets, sg, ln = oracle.es_eval(L, th, noise.noise, idx, 0.02, 12, ref, seeds)
        opt.theta = th.copy()
        _, th = opt.update(oracle.es_gradient(noise.noise, idx, rets, L.P), 0.005)
        th = th.copy()
    assert np.array_equal(theta, th)
    assert [c[0] for c in worker_engine.calls] == ["es_eval", "es_eval"] and worker_engine.calls[0][1] == 4
    assert os.path.exists(os.path.join(str(tmp_path), "log.txt"))


def test_policy_surface(oracle, small_noise, tmp_path):
    from oracle_engine import OracleEngine
    from dne_hip import policies
    eng = OracleEngine(0, ref_count=16)
    eng.noise_upload(small_noise)
    env = policies.HipAtariEnv(eng, seed=3)
    pol = policies.ESAtariPolicy(env.observation_space, env.action_space, engine=eng)
    assert pol.num_params == 1009058 and pol.needs_ref_batch and not pol.needs_ob_stat
    pol.initialize(0)
    ref = oracle.get_ref_batch(seed=0, batch_size=16)
    pol.set_ref_batch([r.astype(np.float32) / np.float32(255.0) for r in ref])     # float obs like the reference Task
    assert np.array_equal(eng.ref, ref)
    rews, t, nov = pol.rollout(env, timestep_limit=15)
    r, s, l, bc = oracle.rollout(oracle.layout(0, 18), pol.get_trainable_flat(), ref, 3, 15, want_bc=True)
    assert t == l and rews.dtype == np.float32 and rews.sum() == r and np.array_equal(nov, bc)
    fn = str(tmp_path / "snap.npz")
    pol.save(fn)
    pol2 = policies.ESAtariPolicy.Load(fn, engine=OracleEngine(0, ref_count=16))
    assert np.array_equal(pol2.get_trainable_flat(), pol.get_trainable_flat())
    # the snapshot holds exactly the reference's variables (policies.py:52-53 over tf.contrib.layers' names, policies.py:319-330)
    arrays = pol.variable_arrays()
    want = ["ESAtariPolicy/" + n + ":0" for n in (
        "conv1/weights", "conv1/biases", "BatchNorm/beta", "BatchNorm/gamma", "BatchNorm/moving_mean", "BatchNorm/moving_variance",
        "conv2/weights", "conv2/biases", "BatchNorm_1/beta", "BatchNorm_1/gamma", "BatchNorm_1/moving_mean", "BatchNorm_1/moving_variance",
        "fc/weights", "fc/biases", "BatchNorm_2/beta", "BatchNorm_2/gamma", "BatchNorm_2/moving_mean", "BatchNorm_2/moving_variance",
        "out/weights", "out/biases")]
    assert sorted(arrays) == sorted(want)
    assert arrays["ESAtariPolicy/conv1/weights:0"].shape == (8, 8, 4, 16) and arrays["ESAtariPolicy/out/weights:0"].shape == (256, 18)
    _, mom = oracle.es_ref_pass_moments(oracle.layout(0, 18), pol.get_trainable_flat(), ref)
    assert np.array_equal(arrays["ESAtariPolicy/BatchNorm/moving_mean:0"], mom[0:16])
    assert np.array_equal(arrays["ESAtariPolicy/BatchNorm_1/moving_variance:0"], mom[64:96])
    assert np.array_equal(arrays["ESAtariPolicy/BatchNorm_2/moving_mean:0"], mom[96:352])
    assert (arrays["ESAtariPolicy/BatchNorm_2/moving_variance:0"] >= 0).all()
    with np.load(fn) as f:
        assert sorted(str(n) for n in f["__variables__"]) == sorted(want) and str(f["__name__"]) == "ESAtariPolicy"
    from dne_hip import h5lite
    if policies.snapshot_extension() == ".h5":     # h5py or libhdf5 present: the reference's own container (policies.py:49-57)
        h5 = str(tmp_path / "snap.h5")
        pol.save(h5)
        pol3 = policies.ESAtariPolicy.Load(h5, engine=OracleEngine(0, ref_count=16))
        assert np.array_equal(pol3.get_trainable_flat(), pol.get_trainable_flat())
        name, (ob_shape, nact), _, got = policies.Policy._read_snapshot(h5)
        assert name == "ESAtariPolicy" and tuple(ob_shape) == (84, 84, 4) and nact == 18 and sorted(got) == sorted(want)
        assert all(np.array_equal(got[k], arrays[k]) and got[k].dtype == np.float32 for k in want)
    else:
        assert not h5lite.available()
        with pytest.raises(RuntimeError):
            pol.save(str(tmp_path / "snap.h5"))    # neither h5py nor libhdf5: the error names the converter
    # initialize_from (policies.py:345-372): a 14-action snapshot seeds the leading columns of an 18-action policy
    small = policies.ESAtariPolicy(env.observation_space, policies._Space(n=14))
    small.initialize(1)
    small.save(str(tmp_path / "small.npz"))
    before = pol.get_trainable_flat()
    pol.initialize_from(str(tmp_path / "small.npz"))
    after, sf = pol.get_trainable_flat(), small.get_trainable_flat()
    o18, o14 = pol.spec["out/weights"][0], small.spec["out/weights"][0]
    assert np.array_equal(after[:o18], sf[:o14])                                       # every layer before the head: taken over
    assert np.array_equal(after[o18:o18 + 256 * 18].reshape(256, 18)[:, :14], sf[o14:o14 + 256 * 14].reshape(256, 14))
    assert np.array_equal(after[o18:o18 + 256 * 18].reshape(256, 18)[:, 14:], before[o18:o18 + 256 * 18].reshape(256, 18)[:, 14:])
    ga = policies.GAAtariPolicy(env.observation_space, env.action_space, nonlin_type="relu", engine=OracleEngine(1))
    assert ga.num_params == 1008450 and not ga.needs_ref_batch
    with pytest.raises(NotImplementedError):
        policies.GAAtariPolicy(env.observation_space, env.action_space, nonlin_type="tanh")


def test_parse_cutoff():
    from dne_hip import es
    assert es.parse_cutoff(5000) == (5000, None, None, 5000, False)
    assert es.parse_cutoff("adaptive:100,0.5,1.5,5000") == (100, 0.5, 1.5, 5000.0, True)
    assert es.parse_cutoff("env_default")[0] is None
    with pytest.raises(NotImplementedError):
        es.parse_cutoff("bogus")


def test_worker_reevaluates_when_the_master_needs_more(oracle, tmp_path):
    """Liveness (es.py:230-265 collects until episodes_per_batch AND timesteps_per_batch): with an odd episodes_per_batch a
    GPU worker's one shard of 2 * (9 // 2) = 8 episodes is not enough; the worker must deliver another shard for the same
    task instead of waiting forever for a new one -- and the master and worker configurations differ, as in a real launch."""
    from oracle_engine import OracleEngine
    from dne_hip import dist, es
    dist.reset_brokers()
    exp = _exp(pop=9, tslimit=6)
    noise = es.SharedNoiseTable(count=2_500_000)
    me, we = OracleEngine(0, ref_count=16), OracleEngine(0, ref_count=16)
    mcfg, rcfg = {"host": "127.0.0.1", "port": 1, "transport": "inprocess"}, {"unix_socket_path": "/tmp/dne_relay_x.sock"}
    out = {}
    tm = threading.Thread(target=lambda: out.update(p=es.run_master(mcfg, str(tmp_path), exp, engine=me, noise=noise, max_iters=1, seed=0)),
                          daemon=True)
    tm.start()
    tw = threading.Thread(target=lambda: es.run_worker(mcfg, rcfg, noise, engine=we, seed=7, reeval_after=0.05), daemon=True)   # max_tasks=None: loops like a reference worker
    tw.start()
    tm.join(timeout=120)
    assert not tm.is_alive(), "master still waiting: the worker never delivered a second shard"
    assert [c[0] for c in we.calls].count("es_eval") >= 2
    assert [c for c in me.calls if c[0] == "es_update"][0][1] >= 8       # the update saw both shards' pairs
