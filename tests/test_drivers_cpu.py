"""CPU: the GA and NS-ES driver mirrors (dne_hip/ga.py, dne_hip/nses.py) over the in-process transport with the
oracle standing in for the GPU engine -- protocol, chain bookkeeping, truncation, novelty plumbing."""
import os
import threading

import numpy as np
import pytest


def _ga_exp(children, parents, tslimit):
    return {"config": {"calc_obstat_prob": 0.0, "episodes_per_batch": children, "eval_prob": 0.0, "l2coeff": 0.005,
                       "noise_stdev": 0.005, "snapshot_freq": 0, "timesteps_per_batch": 10,
                       "return_proc_mode": "centered_rank", "episode_cutoff_mode": tslimit},
            "population_size": parents, "num_elites": 1, "env_id": "FrostbiteNoFrameskip-v4",
            "policy": {"args": {"nonlin_type": "relu"}, "type": "GAAtariPolicy"}}


def test_ga_master_worker(oracle, tmp_path):
    from oracle_engine import OracleEngine
    from dne_hip import dist, es, ga
    dist.reset_brokers()
    exp = _ga_exp(children=6, parents=3, tslimit=25)
    noise = es.SharedNoiseTable(count=2_500_000)
    me, we = OracleEngine(1), OracleEngine(1)
    cfg = {"unix_socket_path": "/tmp/test_ga.sock", "transport": "inprocess"}
    out = {}
    tm = threading.Thread(target=lambda: out.update(r=ga.run_master(cfg, str(tmp_path), exp, engine=me, noise=noise, max_iters=3)), daemon=True)
    tm.start()
    ga.run_worker(cfg, cfg, noise, engine=we, max_tasks=3, seed=11, reeval_after=1e9)
    tm.join(timeout=300)
    assert not tm.is_alive()
    policy, population, score = out["r"]
    assert len(population) == 3 and all(isinstance(c, list) for c in population)
    assert [len(c) for c in population] != [1, 1, 1]          # chains grew over generations (ga.py:252)
    assert np.all(np.diff(score) <= 0)                          # ordered by (-return, arrival)
    # elite theta in slot 0 = normc(noise[s0]) + sigma * sum noise[s_k]   (ga.py:151-158)
    L = oracle.layout(1, 18)
    assert np.array_equal(policy.get_trainable_flat(), oracle.ga_rebuild(L, noise.noise, population[0], 0.005))
    # every chain starts from a generation-0 seed and every child = some parent + one index
    assert all(0 <= s <= noise.noise.size - L.P for c in population for s in c)
    assert [c[0] for c in we.calls] == ["ga_eval"] * 3 and we.calls[0][1] == 6


def test_nses_master_worker(oracle, tmp_path):
    from oracle_engine import OracleEngine
    from dne_hip import dist, es, nses
    dist.reset_brokers()
    exp = {"config": {"calc_obstat_prob": 0.0, "episodes_per_batch": 4, "eval_prob": 0.0, "l2coeff": 0.005,
                      "noise_stdev": 0.02, "snapshot_freq": 0, "timesteps_per_batch": 10,
                      "return_proc_mode": "centered_sign_rank", "episode_cutoff_mode": 10},
           "env_id": "FrostbiteNoFrameskip-v4", "algo_type": "nsr",
           "novelty_search": {"k": 2, "population_size": 2, "num_rollouts": 1, "selection_method": "novelty_prob"},
           "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"}, "policy": {"args": {}, "type": "ESAtariPolicy"}}
    noise = es.SharedNoiseTable(count=2_500_000)
    me, we = OracleEngine(0, ref_count=16, bc_max_steps=10), OracleEngine(0, ref_count=16, bc_max_steps=10)
    cfg = {"unix_socket_path": "/tmp/test_ns.sock", "transport": "inprocess"}
    out = {}
    tm = threading.Thread(target=lambda: out.update(r=nses.run_master(cfg, str(tmp_path), exp, engine=me, noise=noise, max_iters=2)), daemon=True)
    tm.start()
    nses.run_worker(cfg, cfg, noise, engine=we, max_tasks=2, seed=5, reeval_after=1e9)
    tm.join(timeout=300)
    assert not tm.is_alive()
    theta_dict, archive = out["r"]
    assert len(theta_dict) == 2 and len(archive) == 2 + 2      # pop_size initial BCs + one per iteration (nses.py:112,247)
    assert all(a.dtype == np.uint8 and a.shape[1] == 128 and 1 <= a.shape[0] <= 10 for a in archive)
    assert any(not np.array_equal(theta_dict[p], __import__("dne_hip.policies", fromlist=["x"]).xavier_flat(18, p)) for p in (0, 1))


def test_es_modified_vine_dumps(oracle, tmp_path):
    """es_modified.py:140-199: per generation the parent snapshot + reference batch, the parent point and the offspring cloud
    in the reference's row format; every row is reproducible from its (noise_idx, policy_seed, sign) -- the point of the
    policy seeds (policies.py:392-396)."""
    import pickle
    from oracle_engine import OracleEngine
    from dne_hip import dist, es, es_modified, policies
    dist.reset_brokers()
    exp = {"config": {"calc_obstat_prob": 0.0, "episodes_per_batch": 6, "eval_prob": 1.0, "l2coeff": 0.005, "noise_stdev": 0.02,
                      "snapshot_freq": 0, "timesteps_per_batch": 10, "return_proc_mode": "centered_rank", "episode_cutoff_mode": 14},
           "env_id": "FrostbiteNoFrameskip-v4", "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"},
           "policy": {"args": {}, "type": "ESAtariPolicy"}}
    noise = es.SharedNoiseTable(count=2_500_000)
    me, we = OracleEngine(0, ref_count=16), OracleEngine(0, ref_count=16, bc_final_only=True)
    cfg = {"unix_socket_path": "/tmp/test_vine.sock", "transport": "inprocess"}
    root = str(tmp_path / "snapshots")
    out = {}
    tm = threading.Thread(target=lambda: out.update(p=es_modified.run_master(cfg, str(tmp_path), exp, engine=me, noise=noise, max_iters=2,
                                                                            seed=0, snapshot_root=root)), daemon=True)
    tm.start()
    es_modified.run_worker(cfg, cfg, noise, engine=we, max_tasks=2, seed=4, reeval_after=1e9)
    tm.join(timeout=300)
    assert not tm.is_alive()
    L = oracle.layout(0, 18)
    theta0 = policies.xavier_flat(18, 0)
    for gen in range(2):
        d = os.path.join(root, "snapshot_gen_%04d" % gen)
        assert os.path.exists(os.path.join(d, "snapshot_parent_%04d" % gen + policies.snapshot_extension()))
        ref = np.asarray(pickle.load(open(os.path.join(d, "snapshot_parent_%04d_rb.p" % gen), "rb")))
        assert ref.shape == (16, 84, 84, 4)
        cloud = np.loadtxt(os.path.join(d, "snapshot_offspring_%04d.dat" % gen))
        assert cloud.shape == (6, 128 + 5)                                    # 3 pairs x 2 rollouts; RAM, fitness, length, idx, seed, sign
        assert set(cloud[:, -1]) == {1.0, -1.0} and len(set(cloud[cloud[:, -1] == 1, -2])) == 1   # one + seed, one - seed per iteration
        parent = np.loadtxt(os.path.join(d, "snapshot_parent_%04d.dat" % gen))
        assert parent.shape == (128 + 4,) and parent[-1] == 0.02
        if gen == 0:   # replay two offspring rows and the parent row with the oracle
            for row in (cloud[0], cloud[3]):
                th = oracle.perturb(theta0, noise.noise, int(row[-3]), 0.02, int(row[-1]))
                r, s, l, traj = oracle.rollout(L, th, me.ref, int(row[-2]), 14, want_bc=True)
                assert r == row[128] and l == row[129] and np.array_equal(traj[-1], row[:128].astype(np.uint8))
            r, s, l, traj = oracle.rollout(L, theta0, me.ref, int(parent[-2]), 14, want_bc=True)
            assert r == parent[128] and l == parent[129] and np.array_equal(traj[-1], parent[:128].astype(np.uint8))


def test_gpu_tree_ga_schedules_resume_and_elite(oracle, tmp_path):
    """gpu_implementation/ga.py on the engine surface: schedules (helper.py:46-88), genomes with per-seed mutation power,
    validated elite, snapshot.pkl resume -- two iterations in one go equal one iteration + a resumed one."""
    import pickle
    from oracle_engine import OracleEngine
    from dne_hip import es, ga_gpu
    assert ga_gpu.make_schedule(0.002).value(iteration=7) == 0.002
    lin = ga_gpu.make_schedule({"type": "LinearSchedule", "schedule": 10, "initial_p": 0.01, "final_p": 0.001, "field": "iteration"})
    assert lin.value(iteration=0) == 0.01 and abs(lin.value(iteration=5) - 0.0055) < 1e-12 and abs(lin.value(iteration=50) - 0.001) < 1e-15
    ex = ga_gpu.make_schedule({"type": "ExponentialSchedule", "schedule": 4, "initial_p": 0.01, "final_p": 0.0001, "field": "iteration"})
    assert abs(ex.value(iteration=2) - 0.001) < 1e-12 and abs(ex.value(iteration=9) - 0.0001) < 1e-15
    with pytest.raises(AssertionError):
        lin.value(timesteps_so_far=3)
    exp = {"game": "frostbite", "model": "Model", "num_validation_episodes": 2, "num_test_episodes": 3, "population_size": 6,
           "episode_cutoff_mode": 12, "timesteps": 10 ** 9, "validation_threshold": 2, "selection_threshold": 3,
           "mutation_power": {"type": "LinearSchedule", "schedule": 4, "initial_p": 0.004, "final_p": 0.002, "field": "iteration"}}
    noise = es.SharedNoiseTable(count=2_500_000)

    def run(log_dir, iters):
        eng = OracleEngine(1)
        return ga_gpu.main(str(log_dir), engine=eng, noise=noise, seed=5, max_iters=iters, **exp), eng

    (test2, val2, st2), eng2 = run(tmp_path / "a", 2)
    assert st2.it == 2 and len(st2.population) == 6 and st2.elite is not None and val2["val"] == st2.curr_solution_val
    fits = [o.fitness for o in st2.population]
    assert fits == sorted(fits, reverse=True)
    # generation 2's children: a parent genome + one (idx, power) with the scheduled power of iteration 1
    child = next(o for o in st2.population if len(o.seeds) == 2)
    assert isinstance(child.seeds[0], (int, np.integer)) and abs(child.seeds[1][1] - 0.0035) < 1e-12
    assert os.path.exists(tmp_path / "a" / "snapshot.pkl")
    st = pickle.load(open(tmp_path / "a" / "snapshot.pkl", "rb"))
    assert st.it == 2 and [o.seeds for o in st.population] == [o.seeds for o in st2.population]
    # the elite's weights through the engine surface = the reference formula
    th = eng2.ga_rebuild_powers(0, st2.elite.seeds)
    P = eng2.P
    ref = noise.noise[st2.elite.seeds[0]:st2.elite.seeds[0] + P] * ga_gpu.model_scale_by(18)
    for idx, power in st2.elite.seeds[1:]:
        ref = ref + np.float32(power) * noise.noise[idx:idx + P]
    assert np.array_equal(th, ref.astype(np.float32))
    # resume: a second call in the same log_dir continues from iteration 2 with the saved population as parents
    (_, _, st3), eng3 = run(tmp_path / "a", 1)
    assert st3.it == 3 and st3.timesteps_so_far > st2.timesteps_so_far
    assert all(tuple(o.seeds[:-1]) in [tuple(p.seeds) for p in st2.population[:3]] + [tuple(st2.elite.seeds)] for o in st3.population
               if len(o.seeds) > 1 and o.seeds[-1][1] == pytest.approx(0.003))
    sb = ga_gpu.model_scale_by(18)
    assert sb.shape == (1008450,) and sb[0] == np.float32(1 / 16.0) and sb[4096] == 0.0 and abs(sb[-19] - 0.1 / 16.0) < 1e-9


def test_gpu_tree_ga_large_model_on_the_engine_surface(oracle, tmp_path):
    """exp['model'] = 'LargeModel' (configurations/ga_atari_config.json): ga_gpu.main picks engine kind DNE_KIND_GA_LARGE and its
    scale_by (models/dqn.py:25-27 over the LargeModel's shapes); with the oracle behind the engine surface the elite's weights are
    the reference formula noise[idx0] * scale_by + sum power_k * noise[idx_k] (models/base.py:118-149)."""
    from oracle_engine import OracleEngine
    from dne_hip import _lib, es, ga_gpu, policies
    assert ga_gpu.MODEL_KINDS == {"Model": _lib.KIND_GA, "LargeModel": _lib.KIND_GA_LARGE}
    spec, P = policies.flat_layout(_lib.KIND_GA_LARGE, 18)
    assert P == 4052658 and spec["conv3/w"][1] == (3, 3, 64, 64) and spec["fc/w"][1] == (7744, 512)
    sb = ga_gpu.model_scale_by(18, _lib.KIND_GA_LARGE)
    assert sb.shape == (P,) and sb[0] == np.float32(1 / 16.0) and sb[spec["conv1/b"][0]] == 0.0
    assert sb[spec["conv2/w"][0]] == np.float32(1 / np.sqrt(512.0)) and sb[spec["conv3/w"][0]] == np.float32(1 / 24.0)
    assert sb[spec["fc/w"][0]] == np.float32(1 / 88.0) and abs(sb[spec["out/w"][0]] - 0.1 / np.sqrt(512.0)) < 1e-9
    exp = {"game": "frostbite", "model": "LargeModel", "num_validation_episodes": 1, "num_test_episodes": 1, "population_size": 4,
           "episode_cutoff_mode": 5, "timesteps": 10 ** 9, "validation_threshold": 2, "selection_threshold": 2, "mutation_power": 0.002}
    noise = es.SharedNoiseTable(count=4_300_000)
    eng = OracleEngine(_lib.KIND_GA_LARGE)
    test, val, st = ga_gpu.main(str(tmp_path), engine=eng, noise=noise, seed=3, max_iters=2, **exp)
    assert st.it == 2 and len(st.population) == 4 and eng.P == P and np.isfinite(test)
    th = eng.ga_rebuild_powers(0, st.elite.seeds)
    ref = noise.noise[st.elite.seeds[0]:st.elite.seeds[0] + P] * sb
    for idx, power in st.elite.seeds[1:]:
        ref = ref + np.float32(power) * noise.noise[idx:idx + P]
    assert np.array_equal(th, ref.astype(np.float32))


def test_snapshots_with_the_old_schedule_classes_still_load():
    """snapshot.pkl of a run made before ga_gpu's three schedule classes became one: TrainingState.mutation_power unpickles from
    the old class names / attribute dicts into an equivalent Schedule (resume and load_population keep working)"""
    import math
    import pickle
    from dne_hip import ga_gpu

    def old(cls, **attrs):                       # an instance as the old code pickled it: the class by name + its own __dict__
        o = cls.__new__(cls)
        o.__dict__.update(attrs)
        return pickle.loads(pickle.dumps(o))
    c = old(ga_gpu.ConstantSchedule, _value=0.002)
    assert c.value(iteration=5) == 0.002 and c.kind == 'constant'
    lin = old(ga_gpu.LinearSchedule, schedule=100, field='iteration', final_p=0.001, initial_p=0.005)
    assert lin.value(iteration=50, timesteps_so_far=0) == pytest.approx(0.003) and lin.value(iteration=500) == pytest.approx(0.001)
    ex = old(ga_gpu.ExponentialSchedule, initial_p=0.01, final_p=0.0001, schedule=10, field='iteration', linear=object.__new__(object))
    assert ex.value(iteration=5) == pytest.approx(math.exp((math.log(0.01) + math.log(0.0001)) / 2))
    # a whole state: what main() loads on resume
    st = ga_gpu.TrainingState({'mutation_power': 0.002, 'episode_cutoff_mode': 5000})
    st.mutation_power = lin
    st2 = pickle.loads(pickle.dumps(st))
    assert st2.sample(st2.mutation_power) == pytest.approx(0.005) and isinstance(st2.mutation_power, ga_gpu.Schedule)
    # constructing by the old names still works too (experiment files name them)
    assert ga_gpu.LinearSchedule(schedule=10, field='iteration', final_p=1.0, initial_p=0.0).value(iteration=5) == pytest.approx(0.5)
    assert ga_gpu.ConstantSchedule(0.5).value() == 0.5


def _es_gpu_exp(**over):
    exp = {"game": "frostbite", "model": "ModelVirtualBN", "num_validation_episodes": 2, "num_test_episodes": 3, "population_size": 6,
           "timesteps": 10 ** 9, "episode_cutoff_mode": "adaptive:6,0.3,2,20", "return_proc_mode": "centered_rank", "l2coeff": 0.005,
           "mutation_power": {"type": "LinearSchedule", "schedule": 4, "initial_p": 0.02, "final_p": 0.01, "field": "iteration"},
           "optimizer": {"args": {"stepsize": 0.01}, "type": "adam"}}
    exp.update(over)
    return exp


def test_gpu_tree_es_adaptive_cutoff_resume_and_update(oracle, tmp_path):
    """gpu_implementation/es.py on the engine surface: antithetic offspring at the scheduled power, the update of es.py:227-246, the
    adaptive cutoff of es.py:273-276, snapshot.pkl resume (theta AND optimizer state) -- two iterations in one go equal one iteration + a
    resumed one, bit for bit; the first update equals the reference formulas applied to the recorded returns."""
    import pickle
    from oracle_engine import OracleEngine
    from dne_hip import es, es_gpu, policies
    noise = es.SharedNoiseTable(count=2_500_000)
    exp = _es_gpu_exp()

    def run(log_dir, iters, eng=None):
        eng = eng or OracleEngine(0, ref_count=8)
        return es_gpu.main(str(log_dir), engine=eng, noise=noise, seed=4, max_iters=iters, **exp), eng

    st1, e1 = run(tmp_path / "one", 1)
    assert st1.it == 1 and st1.optimizer[2] == 1 and st1.timesteps_so_far > 0 and st1.num_frames == 4 * st1.timesteps_so_far
    # the cutoff grew: every pair's two episodes together reach the 6-step limit (es.py:274 sums a pair's lengths), 100 % > 30 %
    assert st1.tslimit == 12 and st1.adaptive_tslimit and st1.tslimit_max == 20
    # what was evaluated and how it was turned into a step: the engine calls in order (initial test episodes, offspring, update, test episodes)
    kinds = [c[0] for c in e1.calls]
    assert kinds == ["es_eval", "es_eval", "es_update", "es_eval"] and e1.calls[1][1] == 3 and e1.calls[2][1] == 3
    th0 = policies.xavier_flat(18, 4)
    st2, e2 = run(tmp_path / "two", 2)
    assert st2.it == 2 and st2.optimizer[2] == 2
    st1b, _ = run(tmp_path / "one", 1)                       # resumes from snapshot.pkl of the one-iteration run
    assert st1b.it == 2 and st1b.tslimit == st2.tslimit == 20
    assert np.array_equal(st1b.theta, st2.theta) and not np.array_equal(st2.theta, th0)
    for a, b in zip(st1b.optimizer[:2], st2.optimizer[:2]):
        assert np.array_equal(a, b)
    assert st1b.timesteps_so_far == st2.timesteps_so_far
    snap = pickle.load(open(tmp_path / "two" / "snapshot.pkl", "rb"))
    assert snap.it == 2 and np.array_equal(snap.theta, st2.theta) and snap.mutation_power.value(iteration=2) == pytest.approx(0.015)
    log = open(tmp_path / "two" / "log.txt").read()
    for key in ("MutationPower", "TimestepLimitPerEpisode", "PopulationEpRewMedian", "TestRewMean", "InitialRewMax", "TimestepsPerSecondThisIter", "TimestepsComputed"):
        assert key in log
    assert "Increased threshold to 12" in log and "Increased threshold to 20" in log


def test_gpu_tree_es_first_update_is_the_reference_formula(oracle, tmp_path):
    """One iteration of es_gpu.main against the formulas written out with numpy: compute_centered_ranks (es.py:112-121), the weighted sum
    over noise rows divided by 2N (es.py:236-244), Adam on -g + l2coeff * theta (es.py:246, neuroevolution/optimizers.py:55-72)."""
    from oracle_engine import OracleEngine
    from dne_hip import es, es_gpu, policies
    noise = es.SharedNoiseTable(count=2_500_000)
    exp = _es_gpu_exp(episode_cutoff_mode=8, mutation_power=0.02)
    eng = OracleEngine(0, ref_count=8)
    seen = {}
    inner = eng.es_update

    def spy(idx, rets, sg, *a, **k):
        seen["idx"], seen["rets"] = np.array(idx), np.array(rets, np.float32).reshape(-1, 2)
        return inner(idx, rets, sg, *a, **k)
    eng.es_update = spy
    st = es_gpu.main(str(tmp_path), engine=eng, noise=noise, seed=9, max_iters=1, **exp)
    th0 = policies.xavier_flat(18, 9)
    r = seen["rets"]
    ranks = np.empty(r.size, dtype=int); ranks[r.ravel().argsort(kind="stable")] = np.arange(r.size)
    proc = (ranks.reshape(r.shape).astype(np.float32) / (r.size - 1)) - .5
    w = proc[:, 0] - proc[:, 1]
    g = np.dot(w.astype(np.float32), np.stack([noise.noise[i:i + th0.size] for i in seen["idx"]])) / r.size
    grad = -g + 0.005 * th0
    m = 0.1 * grad; v = 0.001 * grad * grad
    a = 0.01 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = th0 + (-a * m / (np.sqrt(v) + 1e-8))
    # Adam's first step is -stepsize * grad / (|grad| + eps'): where |grad| is itself of rounding size the step's sign is not defined,
    # everywhere else the two agree to float32 rounding
    far = np.abs(st.theta - want) > 1e-6
    assert far.mean() < 1e-3 and np.abs(grad[far]).max(initial=0.0) < 1e-5 and np.abs(st.theta - th0).max() > 1e-3
    assert st.tslimit == 8 and not st.adaptive_tslimit


def test_gpu_tree_es_sgd_is_the_engine_sgd_at_a_rescaled_stepsize():
    from dne_hip import es_gpu
    assert es_gpu.engine_optimizer({"type": "sgd", "args": {"stepsize": 0.01, "momentum": 0.9}})[:3] == ("sgd", pytest.approx(0.1), 0.9)
    assert es_gpu.engine_optimizer({"type": "adam", "args": {"stepsize": 0.01}}) == ("adam", 0.01, 0.9, 0.999, 1e-08)
    # v' = mu v + g, step = -lr v'  ==  u' = mu u + (1 - mu) g, step = -(lr / (1 - mu)) u'   with u = (1 - mu) v
    mu, lr, v, u, th_a, th_b = 0.9, 0.01, 0.0, 0.0, 1.0, 1.0
    for g in (0.3, -0.2, 0.5, 0.1):
        v = mu * v + g; th_a -= lr * v
        u = mu * u + (1 - mu) * g; th_b -= lr / (1 - mu) * u
    assert abs(th_a - th_b) < 1e-12
    with pytest.raises(NotImplementedError):
        es_gpu.main("/tmp/_dne_es_gpu_nolog", engine=object(), noise=object(), load_from="x", population_size=2)
