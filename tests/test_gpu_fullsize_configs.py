"""GPU, BASELINE.json configs 3 (deep chains, LargeModel), 4 (NS-ES / NSR-ES) and 5 (the 14-action layout) at the sizes bench.py
times them -- pop 5000 / 1000 children, tslimit 5000, the 250M-entry table -- with EVERY member checked against the CPU oracle
run over the host's usable cores (tests/oracle_pool.py).  The toy-size forms of the same checks live in test_gpu_parity.py /
test_gpu_large.py; test_gpu_fullsize.py holds config 2 and the Deep GA's generations 0 and 1."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

N_PAIRS = 2500
ES_OPT = {"type": "adam", "args": {"stepsize": 0.01}}


def _es_config(**kw):
    from dne_hip import es
    base = dict(l2coeff=0.005, noise_stdev=0.02, episodes_per_batch=5000, timesteps_per_batch=10000, calc_obstat_prob=0.0,
                eval_prob=0.0, snapshot_freq=0, return_proc_mode="centered_rank", episode_cutoff_mode=5000)
    base.update(kw)
    return es.Config(**base)


# ------------------------------------------------------------------------------------------------ config 4
@pytest.mark.timeout(1800)
def test_nses_full_generation_bit_exact(oracle, noise_table, oracle_es_gen0):
    """One NS-ES and one NSR-ES iteration of configurations/frostbite_nses.json's shape at BASELINE size (pop 5000, meta-population
    of 3, k = 10, archive of 32 RAM trajectories, tslimit 5000; nses.py:217-228, 381-384): all 5000 returns and lengths, all 5000
    novelty values (the records' aux slot, Q7) and theta after the rank blend + Adam step, against the oracle.  The rollouts are
    config 2's generation 0 (same theta, reference batch, indices, seeds), so the oracle's trajectories come from the session's
    one run; the archive is built by both sides and compared entry by entry."""
    import oracle_pool
    from dne_hip import _lib, nses as N, policies
    o = oracle_es_gen0
    nact, k, tsl, n_archive, meta_pop = 18, 10, 5000, 32, 3
    cfg = _es_config(return_proc_mode="centered_sign_rank")          # configurations/frostbite_nses.json:10
    e = _lib.Engine(_lib.KIND_ES, nact, max_members=2 * N_PAIRS, ref_count=128, record_bc=True, bc_max_steps=tsl)
    try:
        noise_table.attach(e)
        e.set_ref_batch(o["ref"])
        # nses.py:95-117: the archive starts with the behaviour characterisation of every meta-population member (+ further
        # initialisations, so that the novelty pass sees the archive of a run that is 29 iterations old)
        rs = np.random.RandomState(7)
        thetas = [policies.xavier_flat(nact, seed=100 + p) for p in range(n_archive)]
        bc_seeds = [int(rs.randint(2 ** 31)) for _ in range(n_archive)]
        archive = []
        for th, s in zip(thetas, bc_seeds):
            e.set_theta(th)
            archive.append(N.get_mean_bc(e, tsl, s))
        o_archive = oracle_pool.es_trajectories(thetas, o["ref"], bc_seeds, tsl, nact)
        assert [a.shape for a in archive] == [a.shape for a in o_archive]
        assert all(np.array_equal(a, b) for a, b in zip(archive, o_archive))
        assert len({a.shape[0] for a in archive}) > 1                 # ragged: the padding rule of nses.py:12-20 is exercised
        o_nov = oracle_pool.novelty_all(o_archive, o["bcs"], k).astype(np.float32).reshape(-1, 2)
        o_nov_rank = oracle.centered_ranks(o_nov.reshape(-1)).reshape(-1, 2)
        o_rew_rank = oracle.centered_ranks(o["ret"].reshape(-1)).reshape(-1, 2)
        th0 = o["theta"]
        zeros = np.zeros(e.P, np.float32)
        first = None
        for algo in ("ns", "nsr"):
            e.set_theta(th0); e.optimizer_set_state(zeros, zeros, 0)   # a meta-population member's own (fresh) Adam state
            rec, ratio = N.nses_generation(e, noise_table.noise.size, cfg, algo, archive, k, N_PAIRS, 0, tsl, ES_OPT)
            assert np.array_equal(rec["noise_idx"], o["idx"])
            assert np.array_equal(rec["len"], o["ln"]), np.flatnonzero((rec["len"] != o["ln"]).any(axis=1))[:8]
            assert np.array_equal(rec["ret"], o["ret"])
            assert rec["aux"].dtype == np.float32 and np.array_equal(rec["aux"], o_nov), np.abs(rec["aux"] - o_nov).max()
            proc = o_nov_rank if algo == "ns" else ((o_rew_rank + o_nov_rank) / 2.0).astype(np.float32)   # nses.py:217-228
            g = oracle.weighted_sum(noise_table.noise, o["idx"], proc[:, 0] - proc[:, 1], e.P, float(o["ret"].size))
            oratio, oth = oracle.Adam(th0, 0.01).update(g, cfg.l2coeff)
            assert np.array_equal(e.get_theta(), oth), algo
            assert np.isclose(ratio, oratio, rtol=1e-5)
            m, v, t = e.optimizer_get_state()
            assert t == 1 and m.any() and v.any()
            if first is None:
                first = rec.copy()
            else:
                assert np.array_equal(first, rec)                     # the blend is the only difference between the two
        # the parent's new characterisation joins the archive and the next novelty pass sees 33 entries (nses.py:246-247)
        archive.append(N.get_mean_bc(e, tsl, 12345))
        nov33 = e.novelty_batch(archive, rec["len"], k)
        o_archive.append(oracle_pool.es_trajectories([e.get_theta()], o["ref"], [12345], tsl, nact)[0])
        assert np.array_equal(archive[-1], o_archive[-1])
        some = np.random.RandomState(5).choice(2 * N_PAIRS, 64, replace=False)
        for j in some:
            assert nov33[j] == oracle.novelty(o_archive, o["bcs"][j], k), j
        assert e.check_redzones() == 0
    finally:
        e.close()


# ------------------------------------------------------------------------------------------------ config 5
@pytest.mark.timeout(1800)
def test_fourteen_action_full_generation_bit_exact(oracle, noise_table):
    """Config 5's odd game out: Asteroids has 14 actions (gym_tensorflow/atari/tf_atari.py:158), which moves the output layer and
    everything behind it in the flat vector (P = 1 008 030).  One whole ES generation at pop 5000 / tslimit 5000 through the
    driver's own call (es.es_generation: eval, device-resident exchange, ranks, weighted sum, Adam): all 2500 x 2 returns,
    sign-returns, lengths and theta after the update against the oracle."""
    import oracle_pool
    from dne_hip import _lib, es, policies
    nact, gi = 14, 2                                                   # tools/workloads.py:six_games numbers Asteroids 2
    assert _lib.num_params(_lib.KIND_ES, nact) == 1009058 - 4 * 257
    e = _lib.Engine(_lib.KIND_ES, nact, max_members=2 * N_PAIRS, ref_count=128)
    try:
        noise_table.attach(e)
        th = policies.xavier_flat(nact, seed=gi)
        ref = oracle.get_ref_batch(seed=gi, batch_size=128, nact=nact, env_seed=1000 * gi)
        e.set_theta(th); e.set_ref_batch(ref); e.optimizer_reset()
        gen = 100 * gi
        rec, ratio = es.es_generation(e, noise_table.noise.size, _es_config(), N_PAIRS, gen, 5000, ES_OPT)
        theta_gpu = e.get_theta()
        _, idx, seeds = es.generation_inputs(noise_table.noise.size, e.P, N_PAIRS, gen, 0, 1)
        assert np.array_equal(rec["noise_idx"], idx)
        oret, osg, oln, _ = oracle_pool.es_generation(noise_table.noise, th, ref, idx, seeds, 0.02, 5000, nact)
        assert np.array_equal(rec["len"], oln), np.flatnonzero((rec["len"] != oln).any(axis=1))[:8]
        assert np.array_equal(rec["ret"], oret) and np.array_equal(rec["aux"], osg)
        g = oracle.es_gradient(noise_table.noise, idx, oret, e.P)
        oratio, oth = oracle.Adam(th, 0.01).update(g, 0.005)
        assert np.array_equal(theta_gpu, oth) and np.isclose(ratio, oratio, rtol=1e-5)
        assert e.check_redzones() == 0
    finally:
        e.close()


# ------------------------------------------------------------------------------------------------ config 3, LargeModel
@pytest.mark.timeout(1800)
def test_large_model_full_generation_bit_exact(oracle, noise_table):
    """The GPU tree's protocol on its LargeModel (models/dqn.py:39-47, configurations/ga_atari_config.json) at 1000 children,
    top-20 parents, mutation power 0.002, tslimit 5000 on the 250M table: generation 0's 1000 root genomes pick the parents
    (their 20 best, spot-checked), generation 1 -- 1000 children of those parents, materialised once -- is compared child by
    child (return, sign-return, length) and by its 20 survivors with the oracle."""
    import oracle_pool
    from dne_hip import _lib, ga_gpu
    n, T, power, tsl, nact = 1000, 20, 0.002, 5000, 18
    e = _lib.Engine(_lib.KIND_GA_LARGE, nact, max_members=n)
    try:
        noise_table.attach(e)
        model = ga_gpu.HipModel(e)
        sb = model.scale_by
        L = oracle.layout(oracle.KIND_GA_LARGE, nact)
        rs = np.random.RandomState(0)
        roots = [model.randomize(rs, noise_table) for _ in range(n)]
        seeds0 = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        ret0, sg0, ln0 = e.ga_eval_powers(roots, tsl, seeds0)
        order0 = e.ga_select(ret0, T)
        assert np.array_equal(order0, oracle.ga_select(ret0, T))
        parents = [roots[i] for i in order0]
        parent_theta = [oracle.ga_gpu_rebuild(noise_table.noise, g, sb) for g in parents]
        for j in (0, T - 1):                                           # the parents' own scores, by the oracle
            i = int(order0[j])
            assert (ret0[i], sg0[i], ln0[i]) == oracle.rollout(L, parent_theta[j], None, seeds0[i], tsl)[:3], j
        pick = rs.randint(0, T, size=n)
        kids = [model.mutate(parents[p], rs, noise_table, power) for p in pick]
        seeds1 = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        ret, sg, ln = e.ga_eval_powers(kids, tsl, seeds1)
        fresh = np.array([k[-1][0] for k in kids], np.int64)
        oret, osg, oln = oracle_pool.ga_children(oracle.KIND_GA_LARGE, nact, noise_table.noise, parent_theta, pick, fresh,
                                                 np.full(n, power, np.float32), seeds1, tsl)
        # the pool's child = parent + fl(power * noise[fresh]) is the last step of models/base.py:141-149's chain
        assert np.array_equal(oracle.perturb(parent_theta[pick[0]], noise_table.noise, int(fresh[0]), power, 1),
                              oracle.ga_gpu_rebuild(noise_table.noise, kids[0], sb))
        assert np.array_equal(ln, oln), np.flatnonzero(ln != oln)[:8]
        assert np.array_equal(ret, oret) and np.array_equal(sg, osg)
        assert 1 <= ln.min() and ln.max() <= tsl and len(set(ln.tolist())) > 20
        order = e.ga_select(ret, T)
        assert np.array_equal(order, oracle.ga_select(oret, T)) and ret[order[0]] == oret.max()     # gpu_implementation/ga.py:176
        assert e.check_redzones() == 0
    finally:
        e.close()


# ------------------------------------------------------------------------------------------------ config 3, deep chains
@pytest.mark.timeout(1800)
def test_ga_deep_chain_generation_bit_exact(oracle, noise_table):
    """SURVEY 8d config 3 asks for generations 10 and 100 and the chain of 259 (gpu_implementation/neuroevolution/display.py:31):
    a 20-parent population whose genomes carry 10, 100 and 259 seeds, rebuilt on a COLD parent cache (one streaming pass per
    chain, reduce.h:k_chain_sum) and evaluated with 1000 children at tslimit 5000 -- every rebuilt parent vector, every child's
    return / sign-return / length and the 20 survivors with their scores against the oracle (ga.py:136-149, 251-271)."""
    import oracle_pool
    from dne_hip import _lib, ga
    n, T, sigma, tsl, nact, gen = 1000, 20, 0.005, 5000, 18, 100
    L = oracle.layout(oracle.KIND_GA, nact)
    hi = noise_table.noise.size - L.P + 1
    rs = np.random.RandomState(3)                                      # chains from RandomState(3)
    lengths = [10] * 7 + [100] * 7 + [259] * 6
    population = [[int(s) for s in rs.randint(0, hi, size=m)] for m in lengths]
    scores = np.linspace(60, 0, T).astype(np.float32)                  # the elite's OLD score stays in the contest (ga.py:136-137)
    parent_theta = [oracle.ga_rebuild(L, noise_table.noise, c, sigma) for c in population]
    e = _lib.Engine(_lib.KIND_GA, nact, max_members=n)
    try:
        noise_table.attach(e)
        mine, parent, fresh, env_seeds = ga.ga_generation_inputs(noise_table.noise.size, e.P, n, T, gen, 0, 1)
        chains = [list(population[p]) + [int(f)] for p, f in zip(parent, fresh)]
        assert {len(c) for c in chains} == {11, 101, 260}
        ret, sg, ln = e.ga_eval(chains, sigma, tsl, env_seeds)        # cold cache: the 20 chains are rebuilt inside this call
        for j in (0, 7, 19):                                           # one parent of each chain length, as the engine holds it
            assert np.array_equal(e.ga_rebuild(0, population[j], sigma), parent_theta[j]), j
        oret, osg, oln = oracle_pool.ga_children(oracle.KIND_GA, nact, noise_table.noise, parent_theta, parent, fresh,
                                                 np.full(n, sigma, np.float32), env_seeds, tsl)
        assert np.array_equal(oracle.perturb(parent_theta[parent[0]], noise_table.noise, int(fresh[0]), sigma, 1),
                              oracle.ga_rebuild(L, noise_table.noise, chains[0], sigma))
        assert np.array_equal(ln, oln), np.flatnonzero(ln != oln)[:8]
        assert np.array_equal(ret, oret) and np.array_equal(sg, osg)
        ret2, _, ln2 = e.ga_eval(chains, sigma, tsl, env_seeds)       # warm cache: same bits
        assert np.array_equal(ret2, ret) and np.array_equal(ln2, ln)
        new_pop, new_score, ln3 = ga.ga_generation(e, noise_table.noise.size, sigma, population, scores, n, T, 1, gen, tsl)
        assert np.array_equal(ln3, ln)
        cand = [list(population[0])] + chains
        cand_ret = np.concatenate([scores[:1], oret]).astype(np.float32)
        osel = oracle.ga_select(cand_ret, T)
        assert [cand[i] for i in osel] == [list(c) for c in new_pop] and np.array_equal(cand_ret[osel], new_score)
        assert new_score[0] == cand_ret.max()                                                      # ga.py:149
        assert e.check_redzones() == 0
    finally:
        e.close()
