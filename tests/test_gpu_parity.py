"""GPU parity tests proper: the HIP engine (through the C ABI, ctypes) against the CPU oracle on the same
seeded inputs.  Bit-exact for every integer/byte/index quantity AND for the fp32 forward (the operation
order is part of the contract), 1e-5 against the reference's BLAS-ordered update vector."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NREF = 16
NACT = 18


@pytest.fixture(scope="module")
def hip():
    from dne_hip import _lib
    return _lib


@pytest.fixture(scope="module")
def es_engine(hip, small_noise):
    e = hip.Engine(hip.KIND_ES, NACT, max_members=64, ref_count=NREF, record_bc=True, bc_max_steps=224, profile_events=True)
    e.noise_upload(small_noise)
    yield e
    e.close()


@pytest.fixture(scope="module")
def ga_engine(hip, small_noise):
    e = hip.Engine(hip.KIND_GA, NACT, max_members=64, record_bc=True)
    e.noise_upload(small_noise)
    yield e
    e.close()


@pytest.fixture(scope="module")
def ref_batch(oracle):
    return oracle.get_ref_batch(seed=0, batch_size=NREF, nact=NACT)


def test_library_loaded_is_in_tree(hip):
    import os
    lib = hip.load()
    assert os.path.samefile(lib._name, hip.LIB_PATH)
    assert hip.LIB_PATH.endswith(os.path.join("deep-neuroevolution_amd", "csrc", "libdne_hip.so"))


def test_hip_runtime_is_the_system_one(hip, es_engine):
    """The process that runs the GPU tests must be on /opt/rocm's HIP runtime -- the one libdne_hip.so was built against and
    the one bench.py runs on -- not on the ROCm 7.0 copy bundled with torch (which wins if torch is imported first)."""
    import sys
    maps = open("/proc/self/maps").read()
    hip_libs = sorted({line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line})
    assert hip_libs and all(p.startswith("/opt/rocm") for p in hip_libs), hip_libs
    assert "torch" not in sys.modules, "a test module imported torch at collection time"


def test_noise_get_and_materialize(es_engine, oracle, small_noise):
    e = es_engine
    L = oracle.layout(oracle.KIND_ES, NACT)
    assert e.P == L.P == 1009058
    th = oracle.es_init_theta(L, 0)
    e.set_theta(th)
    assert np.array_equal(e.get_theta(), th)
    assert np.array_equal(e.noise_get(12345, 1000), small_noise[12345:13345])
    idx = np.array([0, 1, 7, 2_990_941, 123_457], np.int64)
    out = e.materialize(idx, 0.02)
    for i, ix in enumerate(idx):
        assert np.array_equal(out[i, 0], oracle.perturb(th, small_noise, ix, 0.02, +1))
        assert np.array_equal(out[i, 1], oracle.perturb(th, small_noise, ix, 0.02, -1))
    # gpu_implementation/es.py:182-183
    assert np.abs((out[:, 0] + out[:, 1]) / 2 - th).max() < 1e-5
    with pytest.raises(Exception):
        e.materialize(np.array([small_noise.size - 10], np.int64), 0.02)


def test_env_reset_and_step(es_engine, oracle):
    e = es_engine
    n = 24
    seeds = np.array([0, 7, 29, 123456789, 5] + list(range(1000, 1000 + n - 5)), np.uint32)
    e.env_reset(seeds)
    envs = [oracle.WrappedEnv() for _ in range(n)]
    obs = np.stack([env.reset(int(s)) for env, s in zip(envs, seeds)])
    assert np.array_equal(e.env_observation(n), obs)
    assert np.array_equal(e.env_ram(n), np.stack([env.ram() for env in envs]))
    rs = np.random.RandomState(0)
    alive = np.ones(n, bool)
    for t in range(230):
        acts = rs.randint(0, NACT, n)
        acts[:6] = 5 if t % 3 else 13          # DOWN / DOWNFIRE pressure: deaths, early done inside a skip
        rew, done = e.env_step(acts)
        g_obs, g_ram = e.env_observation(n), e.env_ram(n)
        for i in range(n):
            if not alive[i]:
                continue
            ob, r, d = envs[i].step(int(acts[i]))
            assert r == rew[i] and d == done[i], (t, i)
            assert np.array_equal(g_ram[i], envs[i].ram()), (t, i)
            assert np.array_equal(g_obs[i], ob), (t, i)
            alive[i] = not d
    assert (~alive).sum() >= 2   # the test really crossed game-over


def test_env_golden_reference_wrappers(es_engine, golden):
    """Same check against the frames produced by the REAL reference wrappers + PIL (tests/golden)."""
    e = es_engine
    seeds = golden["wrap_seeds"]
    e.env_reset(seeds.astype(np.uint32))
    ob = e.env_observation(len(seeds))
    steps = min(len(golden["wrap_s%d_actions" % s]) for s in seeds)
    worst = 0
    for i, s in enumerate(seeds):
        worst = max(worst, np.abs(ob[i].astype(int) - golden["wrap_s%d_obs" % s][0].astype(int)).max())
    for t in range(steps):
        acts = np.array([golden["wrap_s%d_actions" % s][t] for s in seeds], np.int32)
        rew, done = e.env_step(acts)
        ob, ram = e.env_observation(len(seeds)), e.env_ram(len(seeds))
        for i, s in enumerate(seeds):
            assert rew[i] == golden["wrap_s%d_rews" % s][t]
            assert bool(done[i]) == bool(golden["wrap_s%d_dones" % s][t])
            assert np.array_equal(ram[i], golden["wrap_s%d_rams" % s][t])
            worst = max(worst, np.abs(ob[i].astype(int) - golden["wrap_s%d_obs" % s][t + 1].astype(int)).max())
        if done.any():
            break
    assert worst <= 1   # BLAS gray-dot order in the reference: <= 1 LSB


def _members(rs, n, hi):
    off = rs.randint(0, hi, n).astype(np.int64)
    return off


def test_forward_es_bit_exact(es_engine, oracle, small_noise, ref_batch):
    e, O = es_engine, oracle
    L = O.layout(O.KIND_ES, NACT)
    rs = np.random.RandomState(3)
    th = O.es_init_theta(L, 0) + (0.01 * rs.randn(L.P)).astype(np.float32)
    e.set_theta(th)
    e.set_ref_batch(ref_batch)
    n = 10
    off = _members(rs, n, small_noise.size - L.P)
    off[1] = off[0]                      # an antithetic pair
    scale = np.full(n, 0.02, np.float32); scale[1] = -0.02; scale[2] = 0.0; scale[3] = 0.5
    e.set_members(np.zeros(n, np.int32), off, scale)
    obs = rs.randint(0, 256, (n, 84, 84, 4)).astype(np.uint8)
    obs[4] = ref_batch[2]; obs[5] = 0; obs[6] = 255
    e.env_set_observation(obs)
    e.ref_pass(n)
    bn = e.get_bn(n)
    mom = e.get_bn_moments(n)
    acts, logits = e.act(n)
    for i in range(n):
        thi = th + np.float32(scale[i]) * small_noise[off[i]:off[i] + L.P]
        obn, omom = O.es_ref_pass_moments(L, thi, ref_batch)
        assert np.array_equal(bn[i], obn), i
        assert np.array_equal(mom[i], omom), i          # the moving_mean / moving_variance of a snapshot
        y1, y2, y3, lg = O.forward_debug(L, thi, obn, obs[i])
        g1, g2, g3 = e.debug_activations(i)
        assert np.array_equal(g1, y1), i
        assert np.array_equal(g2, y2), i
        assert np.array_equal(g3, y3), i
        assert np.array_equal(logits[i], lg), i
        assert acts[i] == O.act(L, thi, obn, obs[i])[0]


def test_forward_ga_bit_exact(ga_engine, oracle, small_noise):
    e, O = ga_engine, oracle
    L = O.layout(O.KIND_GA, NACT)
    assert e.P == L.P == 1008450
    rs = np.random.RandomState(4)
    parents = [[100], [200_000, 7], [2_900_000, 5, 123_456]]
    ths = []
    for s, chain in enumerate(parents, 1):
        out = e.ga_rebuild(s, chain, 0.005)
        ref = O.ga_rebuild(L, small_noise, chain, 0.005)
        assert np.array_equal(out, ref), chain
        ths.append(ref)
    n = 6
    slot = np.array([1, 2, 3, 1, 2, 3], np.int32)
    off = _members(rs, n, small_noise.size - L.P)
    scale = np.array([0.005, 0.005, 0.005, 0.0, 0.002, 0.005], np.float32)
    e.set_members(slot, off, scale)
    obs = rs.randint(0, 256, (n, 84, 84, 4)).astype(np.uint8)
    e.env_set_observation(obs)
    acts, logits = e.act(n)
    for i in range(n):
        thi = ths[slot[i] - 1] + np.float32(scale[i]) * small_noise[off[i]:off[i] + L.P]
        y1, y2, y3, lg = O.forward_debug(L, thi, None, obs[i])
        g1, g2, g3 = e.debug_activations(i)
        assert np.array_equal(g1, y1) and np.array_equal(g2, y2) and np.array_equal(g3, y3), i
        assert np.array_equal(logits[i], lg), i
        assert acts[i] == int(np.argmax(lg))


def test_es_eval_matches_oracle(es_engine, oracle, small_noise, ref_batch):
    e, O = es_engine, oracle
    L = O.layout(O.KIND_ES, NACT)
    th = O.es_init_theta(L, 0)
    e.set_theta(th)
    e.set_ref_batch(ref_batch)
    n, tslimit, sigma = 8, 220, 0.02
    srs = np.random.RandomState(0)
    idx = np.array([srs.randint(0, small_noise.size - L.P + 1) for _ in range(n)], np.int64)
    seeds = np.random.RandomState(1000).randint(0, 2 ** 31, 2 * n).astype(np.uint32)
    ret, sg, ln, bc = e.es_eval(idx, sigma, tslimit, seeds, want_bc=True)
    oret, osg, oln = O.es_eval(L, th, small_noise, idx, sigma, tslimit, ref_batch, seeds)
    assert np.array_equal(ln, oln)
    assert np.array_equal(ret, oret)
    assert np.array_equal(sg, osg)
    assert ret.dtype == np.float32 and ln.dtype == np.int32 and ret.shape == ln.shape == (n, 2)   # es.py:246-248
    assert ln.min() < tslimit <= ln.max()     # both early game-over and the cutoff were exercised
    # behaviour characterisation (RAM per step) of one member
    thp = O.perturb(th, small_noise, idx[2], sigma, -1)
    r, s, l, obc = O.rollout(L, thp, ref_batch, seeds[5], tslimit, want_bc=True)
    assert l == ln[2, 1] and np.array_equal(bc[5, :l], obc)
    p = e.profile()
    assert p["env_steps"] == ln.sum() and p["eval_ms"] > 0 and p["fc_launches"] == ln.max()
    # eval episode (es.py:388-405): unperturbed theta through the generic member API
    e.set_members(np.zeros(2, np.int32), np.zeros(2, np.int64), np.zeros(2, np.float32))
    r2, s2, l2 = e.eval_members(2, tslimit, seeds[:2])
    for i in range(2):
        orr = O.rollout(L, th, ref_batch, seeds[i], tslimit)
        assert (r2[i], s2[i], l2[i]) == orr[:3]


def test_reduce_and_update(es_engine, oracle, small_noise, golden):
    e, O = es_engine, oracle
    L = O.layout(O.KIND_ES, NACT)
    for key in ("ranks_small_in", "ranks_tied_in", "ranks_distinct_in"):
        x = golden[key]
        assert np.array_equal(e.centered_ranks(x), O.centered_ranks(x.reshape(-1)).reshape(x.shape))
    assert np.array_equal(e.centered_ranks(golden["ranks_distinct_in"]), golden["ranks_distinct_out"])
    n = 96
    srs = np.random.RandomState(0)
    idx = np.array([srs.randint(0, small_noise.size - L.P + 1) for _ in range(n)], np.int64)
    returns = (10 * np.random.RandomState(1).poisson(20, (n, 2))).astype(np.float32)     # SURVEY 8d micro-benchmark
    proc = O.centered_ranks(returns.reshape(-1)).reshape(n, 2)
    w = proc[:, 0] - proc[:, 1]
    g = e.weighted_sum(idx, w, 2 * n)
    og = O.weighted_sum(small_noise, idx, w, L.P, 2 * n)
    assert g.dtype == np.float32 and g.shape == (L.P,)                                    # es.py:297
    assert np.array_equal(g, og)
    # reference formulation (es.py:115-122: chunks of np.dot) within the north star's 1e-5
    ref = np.zeros(L.P, np.float32)
    for s in range(0, n, 500):
        ref += np.dot(w[s:s + 500], np.stack([small_noise[i:i + L.P] for i in idx[s:s + 500]]))
    ref /= returns.size
    assert np.abs(g - ref).max() < 1e-5
    th0 = O.es_init_theta(L, 0)
    for kind, mk in (("adam", lambda: O.Adam(th0, 0.01)), ("sgd", lambda: O.SGD(th0, 0.01, 0.9))):
        e.set_theta(th0); e.optimizer_reset()
        opt = mk()
        for it in range(3):
            rets = (10 * np.random.RandomState(5 + it).poisson(20, (n, 2))).astype(np.float32)
            ratio = e.es_update(idx, rets, None, "centered_rank", kind, 0.005, 0.01)
            og = O.es_gradient(small_noise, idx, rets, L.P)
            oratio, oth = opt.update(og, 0.005)
            assert np.array_equal(e.get_theta(), oth), (kind, it)
            assert abs(ratio - oratio) <= 1e-9 * oratio


def test_ga_eval_and_select(ga_engine, oracle, small_noise):
    e, O = ga_engine, oracle
    L = O.layout(O.KIND_GA, NACT)
    sigma, tslimit = 0.005, 60
    rs = np.random.RandomState(7)
    hi = small_noise.size - L.P + 1
    # generation 0: no parents (ga.py:253-254) ; generation 1: children of two parents
    gen0 = [[int(rs.randint(hi))] for _ in range(6)]
    seeds = rs.randint(0, 2 ** 31, 6).astype(np.uint32)
    ret, sg, ln, bc = e.ga_eval(gen0, sigma, tslimit, seeds, want_bc=True)
    for i, chain in enumerate(gen0):
        th = O.ga_rebuild(L, small_noise, chain, sigma)
        r, s, l, obc = O.rollout(L, th, None, seeds[i], tslimit, want_bc=True)
        assert (ret[i], sg[i], ln[i]) == (r, s, l), i
        assert np.array_equal(bc[i], obc)
    parents = [gen0[1], gen0[4]]
    gen1 = [parents[i % 2] + [int(rs.randint(hi))] for i in range(6)]
    gen1[5] = gen1[5] + [int(rs.randint(hi))]       # a longer chain whose prefix is not cached
    seeds = rs.randint(0, 2 ** 31, 6).astype(np.uint32)
    ret, sg, ln = e.ga_eval(gen1, sigma, tslimit, seeds)
    for i, chain in enumerate(gen1):
        th = O.ga_rebuild(L, small_noise, chain, sigma)
        assert (ret[i], sg[i], ln[i]) == O.rollout(L, th, None, seeds[i], tslimit)[:3], i
    scores = np.array([30, 10, 30, 50, 10, 0, 50, 30], np.float32)
    sel = e.ga_select(scores, 4)
    assert np.array_equal(sel, O.ga_select(scores, 4)) and sel.tolist() == [3, 6, 0, 2]
    assert scores[sel[0]] == scores.max()                                                  # ga.py:149


def test_novelty(es_engine, oracle, golden):
    lens = golden["nov_lens"]
    arch, o = [], 0
    for n in lens:
        arch.append(golden["nov_arch"][o:o + n]); o += n
    bc = golden["nov_bc"]
    for k, key in ((10, "nov_k10"), (3, "nov_k3")):
        v = es_engine.novelty(arch, bc, k)
        assert v == oracle.novelty(arch, bc, k)
        assert np.isclose(v, float(golden[key]), rtol=1e-13)


def test_errors_are_loud(es_engine, hip):
    e = es_engine
    with pytest.raises(hip.DneError):
        e.set_theta(np.zeros(10, np.float32))
    with pytest.raises(hip.DneError):
        e.es_eval(np.array([2 ** 40], np.int64), 0.02, 10, np.zeros(2, np.uint32))
    with pytest.raises(hip.DneError):
        e.env_step(np.array([99], np.int32))
    with pytest.raises(hip.DneError):
        e.env_reset(np.zeros(10 ** 6, np.uint32))


def test_novelty_batch_and_optimizer_state(es_engine, oracle, small_noise, ref_batch):
    e, O = es_engine, oracle
    L = O.layout(O.KIND_ES, NACT)
    th = O.es_init_theta(L, 0)
    e.set_theta(th)
    e.set_ref_batch(ref_batch)
    n, tslimit, sigma = 3, 40, 0.02
    idx = np.array([11, 500_000, 2_000_000], np.int64)
    seeds = np.arange(6, dtype=np.uint32) + 77
    ret, sg, ln = e.es_eval(idx, sigma, tslimit, seeds)          # trajectories stay on the device
    rs = np.random.RandomState(2)
    archive = [rs.randint(0, 256, (m, 128)).astype(np.uint8) for m in (40, 7, 55, 40, 23)]
    nov = e.novelty_batch(archive, ln, 3)
    for i in range(n):
        for s in range(2):
            thp = O.perturb(th, small_noise, idx[i], sigma, 1 if s == 0 else -1)
            _, _, l, bc = O.rollout(L, thp, ref_batch, seeds[2 * i + s], tslimit, want_bc=True)
            assert l == ln[i, s]
            assert nov[2 * i + s] == O.novelty(archive, bc, 3)
    # optimizer state round trip (nses.py keeps one optimizer per meta-population member)
    e.optimizer_reset()
    g_idx = np.array([5, 9], np.int64)
    e.es_update(g_idx, ret[:2], None, "centered_rank", "adam", 0.005, 0.01)
    m, v, t = e.optimizer_get_state()
    th1 = e.get_theta()
    assert t == 1 and m.any() and v.any()
    e.es_update(g_idx, ret[:2], None, "centered_rank", "adam", 0.005, 0.01)
    th2 = e.get_theta()
    e.set_theta(th1); e.optimizer_set_state(m, v, t)
    e.es_update(g_idx, ret[:2], None, "centered_rank", "adam", 0.005, 0.01)
    assert np.array_equal(e.get_theta(), th2)


def test_ga_driver_on_device(hip, oracle, small_noise, tmp_path):
    """dne_hip.ga master/worker over the real engine for two generations; elite theta checked against the oracle."""
    import threading
    from dne_hip import dist, es, ga
    dist.reset_brokers()
    exp = {"config": {"calc_obstat_prob": 0.0, "episodes_per_batch": 12, "eval_prob": 0.0, "l2coeff": 0.005,
                      "noise_stdev": 0.005, "snapshot_freq": 0, "timesteps_per_batch": 10,
                      "return_proc_mode": "centered_rank", "episode_cutoff_mode": 30},
           "population_size": 4, "num_elites": 1, "env_id": "FrostbiteNoFrameskip-v4",
           "policy": {"args": {"nonlin_type": "relu"}, "type": "GAAtariPolicy"}}
    noise = es.SharedNoiseTable(count=small_noise.size)
    me = hip.Engine(hip.KIND_GA, NACT, max_members=16)
    we = hip.Engine(hip.KIND_GA, NACT, max_members=16)
    cfg = {"unix_socket_path": "/tmp/gpu_ga.sock", "transport": "inprocess"}
    out = {}
    tm = threading.Thread(target=lambda: out.update(r=ga.run_master(cfg, str(tmp_path), exp, engine=me, noise=noise, max_iters=2)), daemon=True)
    tm.start()
    ga.run_worker(cfg, cfg, noise, engine=we, max_tasks=2, seed=3, reeval_after=1e9)
    tm.join(timeout=120)
    assert not tm.is_alive()
    policy, population, score = out["r"]
    L = oracle.layout(oracle.KIND_GA, NACT)
    assert np.array_equal(policy.get_trainable_flat(), oracle.ga_rebuild(L, small_noise, population[0], 0.005))
    # the elite's score is reproducible by the oracle with the same env seed stream
    rs = np.random.RandomState(3); rs.randint(2 ** 31)
    hi = small_noise.size - L.P + 1
    gen0 = [[int(rs.randint(0, hi))] for _ in range(12)]
    seeds0 = rs.randint(0, 2 ** 32, size=12, dtype=np.uint64).astype(np.uint32)
    r0 = np.array([oracle.rollout(L, oracle.ga_rebuild(L, small_noise, c, 0.005), None, seeds0[i], 30)[0] for i, c in enumerate(gen0)], np.float32)
    sel = oracle.ga_select(r0, 4)
    assert score.max() >= r0[sel[0]]                                # elites are never lost (ga.py:136-137)
    me.close(); we.close()


def test_ref_pass_full_reference_batch(hip, oracle, small_noise):
    """The 128-frame virtual-batch-norm pass of the real configuration (MT = 8 matrix-core path)."""
    O = oracle
    L = O.layout(O.KIND_ES, NACT)
    ref = O.get_ref_batch(seed=0, batch_size=128, nact=NACT)
    e = hip.Engine(hip.KIND_ES, NACT, max_members=24, ref_count=128, ref_chunk=16)
    e.noise_upload(small_noise)
    th = O.es_init_theta(L, 1)
    e.set_theta(th)
    e.set_ref_batch(ref)
    n = 20                       # two chunks (16 + 4) and a partially filled XCD group
    rs = np.random.RandomState(8)
    off = rs.randint(0, small_noise.size - L.P, n).astype(np.int64)
    scale = np.where(np.arange(n) % 2 == 0, 0.02, -0.02).astype(np.float32)
    e.set_members(np.zeros(n, np.int32), off, scale)
    e.ref_pass(n)
    bn = e.get_bn(n)
    for i in (0, 1, 7, 15, 16, 19):
        thi = th + np.float32(scale[i]) * small_noise[off[i]:off[i] + L.P]
        assert np.array_equal(bn[i], O.es_ref_pass(L, thi, ref)), i
    e.close()


def test_nses_driver_on_device(hip, oracle, small_noise, tmp_path):
    """dne_hip.nses master/worker over the real engine (NSR-ES, 2 meta-population members, 2 iterations): the novelty
    shipped in signreturns equals the oracle's novelty of the same rollouts against the same archive."""
    import threading
    from dne_hip import dist, es, nses
    dist.reset_brokers()
    tsl = 24
    exp = {"config": {"calc_obstat_prob": 0.0, "episodes_per_batch": 8, "eval_prob": 0.0, "l2coeff": 0.005,
                      "noise_stdev": 0.02, "snapshot_freq": 0, "timesteps_per_batch": 10,
                      "return_proc_mode": "centered_sign_rank", "episode_cutoff_mode": tsl},
           "env_id": "FrostbiteNoFrameskip-v4", "algo_type": "nsr",
           "novelty_search": {"k": 2, "population_size": 2, "num_rollouts": 1, "selection_method": "round_robin"},
           "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"}, "policy": {"args": {}, "type": "ESAtariPolicy"}}
    noise = es.SharedNoiseTable(count=small_noise.size)
    mk = lambda: hip.Engine(hip.KIND_ES, NACT, max_members=8, ref_count=NREF, record_bc=True, bc_max_steps=tsl)
    me, we = mk(), mk()
    cfg = {"unix_socket_path": "/tmp/gpu_ns.sock", "transport": "inprocess"}
    out, pushed = {}, []
    orig_push = dist.WorkerClient.push_result

    def spy(self, task_id, result):
        pushed.append((task_id, result, self.get_archive()))
        return orig_push(self, task_id, result)

    dist.WorkerClient.push_result = spy
    try:
        tm = threading.Thread(target=lambda: out.update(r=nses.run_master(cfg, str(tmp_path), exp, engine=me, noise=noise, max_iters=2)), daemon=True)
        tm.start()
        nses.run_worker(cfg, cfg, noise, engine=we, max_tasks=2, seed=9, reeval_after=1e9)
        tm.join(timeout=120)
    finally:
        dist.WorkerClient.push_result = orig_push
    assert not tm.is_alive()
    theta_dict, archive = out["r"]
    assert len(archive) == 4 and all(a.shape[1] == 128 and a.dtype == np.uint8 for a in archive)
    # re-derive the first task's novelty with the oracle
    task_id, res, arch = pushed[0]
    L = oracle.layout(oracle.KIND_ES, NACT)
    from dne_hip import policies
    th = policies.xavier_flat(NACT, 0)
    ref = me.env_observation  # noqa: F841  (reference batch lives on the device; rebuild it the same way)
    env = policies.HipAtariEnv(me, seed=0)
    refb = np.rint(np.stack(es.get_ref_batch(env, NREF, np.random.RandomState(0))) * 255.0).astype(np.uint8)
    rs = np.random.RandomState(9); rs.randint(2 ** 31)
    idx = np.sort(np.array([noise.sample_index(rs, L.P) for _ in range(4)], np.int64))   # the worker labels its draws in table order
    seeds = rs.randint(0, 2 ** 32, size=8, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(res.noise_inds_n, idx)
    for i in range(4):
        for s in range(2):
            thp = oracle.perturb(th, small_noise, idx[i], 0.02, 1 if s == 0 else -1)
            r, _, l, bc = oracle.rollout(L, thp, refb, seeds[2 * i + s], tsl, want_bc=True)
            assert r == res.returns_n2[i, s] and l == res.lengths_n2[i, s]
            assert np.float32(oracle.novelty(arch, bc, 2)) == res.signreturns_n2[i, s]
    me.close(); we.close()


def test_fourteen_actions_bit_exact(hip, oracle, small_noise):
    """Asteroids has 14 actions (gym_tensorflow/atari/tf_atari.py:158): the out layer's width moves every offset behind it in
    the flat vector.  ES evaluation + update and a GA evaluation at n_actions = 14, bit-exact against the oracle."""
    O, nact = oracle, 14
    L = O.layout(O.KIND_ES, nact)
    assert L.P == hip.num_params(hip.KIND_ES, nact) == 1009058 - 4 * 257
    ref = O.get_ref_batch(seed=0, batch_size=NREF, nact=nact)
    e = hip.Engine(hip.KIND_ES, nact, max_members=16, ref_count=NREF)
    try:
        e.noise_upload(small_noise)
        th = O.es_init_theta(L, 0)
        e.set_theta(th); e.set_ref_batch(ref)
        n, tslimit = 7, 150
        idx = np.random.RandomState(2).randint(0, small_noise.size - L.P + 1, n).astype(np.int64)
        seeds = np.random.RandomState(3).randint(0, 2 ** 31, 2 * n).astype(np.uint32)
        ret, sg, ln = e.es_eval(idx, 0.02, tslimit, seeds)
        oret, osg, oln = O.es_eval(L, th, small_noise, idx, 0.02, tslimit, ref, seeds)
        assert np.array_equal(ln, oln) and np.array_equal(ret, oret) and np.array_equal(sg, osg)
        rec = e.allgather_results(n, n)
        assert np.array_equal(rec["ret"], oret) and np.array_equal(rec["aux"], osg) and np.array_equal(rec["noise_idx"], idx)
        e.es_update_gathered("centered_rank", "adam", 0.005, 0.01)
        _, oth = O.Adam(th, 0.01).update(O.es_gradient(small_noise, idx, oret, L.P), 0.005)
        assert np.array_equal(e.get_theta(), oth)
        # explicit members: logits of the 14-wide head
        e.set_theta(th)
        e.set_members(np.zeros(3, np.int32), idx[:3], np.array([0.02, -0.02, 0.0], np.float32))
        obs = np.random.RandomState(4).randint(0, 256, (3, 84, 84, 4)).astype(np.uint8)
        e.env_set_observation(obs)
        e.ref_pass(3)
        acts, logits = e.act(3)
        assert logits.shape == (3, 14)
        for i, sc in enumerate((0.02, -0.02, 0.0)):
            thi = th + np.float32(sc) * small_noise[idx[i]:idx[i] + L.P]
            bn = O.es_ref_pass(L, thi, ref)
            a, lg = O.act(L, thi, bn, obs[i])
            assert np.array_equal(logits[i], lg) and acts[i] == a
        assert e.check_redzones() == 0
    finally:
        e.close()
    Lg = O.layout(O.KIND_GA, nact)
    g = hip.Engine(hip.KIND_GA, nact, max_members=8)
    try:
        g.noise_upload(small_noise)
        chains = [[5], [77, 123_456], [2_000_000, 9, 31]]
        seeds = np.array([11, 12, 13], np.uint32)
        ret, sg, ln = g.ga_eval(chains, 0.005, 80, seeds)
        for i, c in enumerate(chains):
            r = O.rollout(Lg, O.ga_rebuild(Lg, small_noise, c, 0.005), None, seeds[i], 80)
            assert (ret[i], sg[i], ln[i]) == r[:3], i
        assert g.check_redzones() == 0
    finally:
        g.close()


def test_exchange_records_roundtrip(es_engine, oracle, small_noise, ref_batch):
    """The exchange surface on one GPU: pack (this rank's shard as 32-byte wire records) -> set (gathered records from the
    host) -> update equals the all-in-one device path and the host-array entry point, bit for bit; a one-rank RCCL
    communicator built through dne_comm_* runs the same all-gather through librccl."""
    e, O = es_engine, oracle
    L = O.layout(O.KIND_ES, NACT)
    th = O.es_init_theta(L, 0)
    e.set_ref_batch(ref_batch)
    n = 6
    idx = np.random.RandomState(5).randint(0, small_noise.size - L.P + 1, n).astype(np.int64)
    seeds = np.random.RandomState(6).randint(0, 2 ** 31, 2 * n).astype(np.uint32)
    thetas = []
    for path in ("gathered", "pack_set", "host"):
        e.set_theta(th); e.optimizer_reset()
        ret, sg, ln = e.es_eval(idx, 0.02, 30, seeds)
        if path == "gathered":
            rec = e.allgather_results(n, n)
            assert rec.dtype.itemsize == 32
            assert np.array_equal(rec["noise_idx"], idx) and np.array_equal(rec["ret"], ret)
            assert np.array_equal(rec["len"], ln) and np.array_equal(rec["aux"], sg)
            e.es_update_gathered("centered_rank", "adam", 0.005, 0.01)
        elif path == "pack_set":
            rec2 = e.records_pack(n)
            assert rec2.tobytes() == rec.tobytes()
            e.records_set(rec2)
            e.es_update_gathered("centered_rank", "adam", 0.005, 0.01)
        else:
            e.es_update(idx, ret, sg, "centered_rank", "adam", 0.005, 0.01)
        thetas.append(e.get_theta())
    assert np.array_equal(thetas[0], thetas[1]) and np.array_equal(thetas[0], thetas[2])
    _, oth = O.Adam(th, 0.01).update(O.es_gradient(small_noise, idx, ret, L.P), 0.005)
    assert np.array_equal(thetas[0], oth)
    with pytest.raises(Exception):
        e.allgather_results(n - 1, n)          # a shard that does not match the population / rank layout
    bad = rec.copy(); bad["noise_idx"][2] = small_noise.size
    with pytest.raises(Exception):
        e.records_set(bad)                     # a noise index outside the table never reaches the aggregate kernel


@pytest.mark.parametrize("n_global,world", [(7, 2), (2500, 8), (2500, 3), (5, 8), (64, 4)])
def test_unshard_matches_the_host_transport(es_engine, n_global, world):
    """The device-side un-sharding of an all-gather result (what dne_allgather_results runs after ncclAllGather) against the
    host transport's reordering (es.allgather_records): ragged shards, more ranks than pairs."""
    from dne_hip import es
    e = es_engine
    rs = np.random.RandomState(n_global * 31 + world)
    full = np.zeros(n_global, es.RECORD)
    full["noise_idx"] = rs.randint(0, 2 ** 40, n_global)
    full["ret"] = rs.randn(n_global, 2).astype(np.float32); full["aux"] = rs.randn(n_global, 2).astype(np.float32)
    full["len"] = rs.randint(1, 5000, (n_global, 2))
    per = (n_global + world - 1) // world
    gathered = np.zeros((world, per), es.RECORD)
    for r in range(world):
        ids = es.shard_pairs(n_global, r, world)
        gathered[r, :len(ids)] = full[ids]
    assert e.debug_unshard(gathered, n_global, world).tobytes() == full.tobytes()


def test_rccl_single_rank_comm(hip, oracle, small_noise, ref_batch):
    O = oracle
    L = O.layout(O.KIND_ES, NACT)
    e = hip.Engine(hip.KIND_ES, NACT, max_members=16, ref_count=NREF)
    try:
        uid = hip.comm_unique_id()
        assert len(uid) == 128
        e.comm_init(0, 1, uid)
        e.barrier()
        assert e.comm_allreduce([1.5, 2.0], "sum").tolist() == [1.5, 2.0]
        assert e.comm_allreduce([3.0], "max").tolist() == [3.0]
        e.noise_upload(small_noise)
        th = O.es_init_theta(L, 0)
        e.set_theta(th); e.set_ref_batch(ref_batch)
        n = 5
        idx = np.random.RandomState(8).randint(0, small_noise.size - L.P + 1, n).astype(np.int64)
        seeds = np.arange(2 * n, dtype=np.uint32) + 50
        ret, sg, ln = e.es_eval(idx, 0.02, 20, seeds)
        rec = e.allgather_results(n, n)
        assert np.array_equal(rec["ret"], ret) and np.array_equal(rec["len"], ln) and np.array_equal(rec["noise_idx"], idx)
        e.es_update_gathered("centered_rank", "adam", 0.005, 0.01)
        _, oth = O.Adam(th, 0.01).update(O.es_gradient(small_noise, idx, ret, L.P), 0.005)
        assert np.array_equal(e.get_theta(), oth)
    finally:
        e.close()


def test_debug_sync_and_trace_knobs(hip, oracle, small_noise, ref_batch, monkeypatch, capfd):
    """DNE_DEBUG_SYNC=1 / DNE_TRACE=1: same results, with a synchronise + error check after every launch set and stage
    breadcrumbs on stderr (what one turns on to localise a device fault)."""
    O = oracle
    L = O.layout(O.KIND_ES, NACT)
    monkeypatch.setenv("DNE_DEBUG_SYNC", "1"); monkeypatch.setenv("DNE_TRACE", "1")
    e = hip.Engine(hip.KIND_ES, NACT, max_members=16, ref_count=NREF)
    try:
        e.noise_upload(small_noise)
        th = O.es_init_theta(L, 0)
        e.set_theta(th); e.set_ref_batch(ref_batch)
        idx = np.array([9, 99_999, 1_234_567], np.int64)
        seeds = np.arange(6, dtype=np.uint32) + 7
        ret, sg, ln = e.es_eval(idx, 0.02, 25, seeds)
        oret, osg, oln = O.es_eval(L, th, small_noise, idx, 0.02, 25, ref_batch, seeds)
        assert np.array_equal(ret, oret) and np.array_equal(ln, oln)
    finally:
        e.close()
    err = capfd.readouterr().err
    assert "engine created" in err and "reference pass done" in err and "lock-step" in err


def test_es_final_ram_only(hip, oracle, small_noise, ref_batch):
    """bc_final_only engines (es_modified.py's dumps use bc_vec[-1]): the final RAM of every member, policy-seeded rollouts."""
    O = oracle
    L = O.layout(O.KIND_ES, NACT)
    e = hip.Engine(hip.KIND_ES, NACT, max_members=16, ref_count=NREF, record_bc=True, bc_final_only=True)
    try:
        e.noise_upload(small_noise)
        th = O.es_init_theta(L, 0)
        e.set_theta(th); e.set_ref_batch(ref_batch)
        idx = np.array([3, 500_000, 2_000_000], np.int64)
        seeds = np.tile(np.array([777, 888], np.uint32), 3)          # one policy seed for all + rollouts, one for all -
        ret, sg, ln, bc = e.es_eval(idx, 0.02, 60, seeds, want_bc=True)
        assert bc.shape == (6, 128)
        for i in range(3):
            for s, sign in enumerate((1, -1)):
                r, q, l, traj = O.rollout(L, O.perturb(th, small_noise, idx[i], 0.02, sign), ref_batch, seeds[2 * i + s], 60, want_bc=True)
                assert (r, q, l) == (ret[i, s], sg[i, s], ln[i, s]) and np.array_equal(bc[2 * i + s], traj[-1])
        e.set_members(np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.float32))
        r1, _, l1, b1 = e.eval_members(1, 60, np.array([4242], np.uint32), want_bc=True)
        r2, _, l2, b2 = e.eval_members(1, 60, np.array([4242], np.uint32), want_bc=True)
        assert (r1, l1) == (r2, l2) and np.array_equal(b1, b2) and b1.shape == (1, 128)      # same policy seed -> same episode
        with pytest.raises(Exception):
            e.novelty_batch([np.zeros((3, 128), np.uint8)], ln, 1)                           # needs full trajectories
    finally:
        e.close()


def test_gpu_tree_genomes_bit_exact(ga_engine, oracle, small_noise):
    """Genomes of the reference's GPU tree -- ((idx0,), (idx1, power1), ...), root = noise[idx0] * scale_by, one mutation power
    per seed (gpu_implementation/neuroevolution/models/base.py:118-149) -- rebuilt and evaluated on the device."""
    from dne_hip import ga_gpu
    e, O = ga_engine, oracle
    L = O.layout(O.KIND_GA, NACT)
    sb = ga_gpu.model_scale_by(NACT)
    e.ga_set_init_scale(sb)
    genomes = [(1234,), (200_000, (7, 0.002)), (2_900_000, (5, 0.004), (123_456, 0.001), (99, 0.0005)),
               (200_000, (7, 0.002), (31_337, 0.003))]
    for g in genomes:
        assert np.array_equal(e.ga_rebuild_powers(1, g), O.ga_gpu_rebuild(small_noise, g, sb)), g
    seeds = np.array([21, 22, 23, 24], np.uint32)
    ret, sg, ln = e.ga_eval_powers(genomes, 70, seeds)
    for i, g in enumerate(genomes):
        r = O.rollout(L, O.ga_gpu_rebuild(small_noise, g, sb), None, seeds[i], 70)
        assert (ret[i], sg[i], ln[i]) == r[:3], i
    # children of the cached parents (the last mutation applied on the fly), then a plain-sigma generation on the same engine:
    # the parent cache must not mix the two genome forms
    kids = [genomes[1] + ((555, 0.0025),), genomes[3] + ((777_777, 0.0015),), genomes[1] + ((556, 0.0025),)]
    ret, sg, ln = e.ga_eval_powers(kids, 70, seeds[:3])
    for i, g in enumerate(kids):
        r = O.rollout(L, O.ga_gpu_rebuild(small_noise, g, sb), None, seeds[i], 70)
        assert (ret[i], sg[i], ln[i]) == r[:3], i
    chains = [[200_000, 7], [200_000]]
    ret, sg, ln = e.ga_eval(chains, 0.002, 50, seeds[:2])
    for i, c in enumerate(chains):
        r = O.rollout(L, O.ga_rebuild(L, small_noise, c, 0.002), None, seeds[i], 50)
        assert (ret[i], sg[i], ln[i]) == r[:3], i


def test_gpu_tree_es_driver_equals_the_oracle_engine(hip, oracle, tmp_path):
    """dne_hip/es_gpu.py (gpu_implementation/es.py: scheduled mutation power, adaptive cutoff, test episodes of the unperturbed theta as
    pairs at power 0, snapshot.pkl) on the HIP engine against the same driver on the oracle behind the same method surface: two
    iterations, theta and Adam's state bit for bit, every counter equal."""
    from oracle_engine import OracleEngine
    from dne_hip import es, es_gpu
    noise = es.SharedNoiseTable(count=2_500_000)
    exp = {"game": "frostbite", "model": "ModelVirtualBN", "num_test_episodes": 3, "population_size": 8, "timesteps": 10 ** 9,
           "episode_cutoff_mode": "adaptive:10,0.3,2,40", "return_proc_mode": "centered_rank", "l2coeff": 0.005,
           "mutation_power": {"type": "LinearSchedule", "schedule": 4, "initial_p": 0.02, "final_p": 0.01, "field": "iteration"},
           "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"}}
    e = hip.Engine(hip.KIND_ES, NACT, max_members=8, ref_count=NREF)
    try:
        sg = es_gpu.main(str(tmp_path / "gpu"), engine=e, noise=noise, seed=2, max_iters=2, **exp)
    finally:
        e.close()
    so = es_gpu.main(str(tmp_path / "cpu"), engine=OracleEngine(0, ref_count=NREF, max_members=8), noise=noise, seed=2, max_iters=2, **exp)
    assert sg.it == so.it == 2 and sg.tslimit == so.tslimit and sg.timesteps_so_far == so.timesteps_so_far and sg.num_frames == so.num_frames
    assert np.array_equal(sg.theta, so.theta)
    assert sg.optimizer[2] == so.optimizer[2] == 2 and np.array_equal(sg.optimizer[0], so.optimizer[0]) and np.array_equal(sg.optimizer[1], so.optimizer[1])
