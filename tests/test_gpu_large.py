"""GPU: the GPU tree's LargeModel (gpu_implementation/neuroevolution/models/dqn.py:39-47; the model configurations/ga_atari_config.json
names) on the HIP engine (DNE_KIND_GA_LARGE, csrc/forward_large.h) against the oracle's restatement -- every layer's raw output, the
logits and the action bit for bit, genomes with per-seed powers rebuilt and evaluated, and the Deep-GA driver of ga_gpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NACT = 18


@pytest.fixture(scope="module")
def big_noise():
    """first 9M entries of the reference noise stream (es.py:60, seed 123): a LargeModel slice is 4.05M floats"""
    return np.random.RandomState(123).randn(9_000_000).astype(np.float32)


@pytest.fixture(scope="module")
def large_engine(big_noise):
    from dne_hip import _lib, ga_gpu
    e = _lib.Engine(_lib.KIND_GA_LARGE, NACT, max_members=16, record_bc=True)
    e.noise_upload(big_noise)
    e.ga_set_init_scale(ga_gpu.model_scale_by(NACT, _lib.KIND_GA_LARGE))
    yield e
    e.close()


def test_num_params_and_layout(oracle):
    from dne_hip import _lib, policies
    spec, P = policies.flat_layout(_lib.KIND_GA_LARGE, NACT)
    L = oracle.layout(oracle.KIND_GA_LARGE, NACT)
    assert P == L.P == 4052658 and _lib.load().dne_num_params(_lib.KIND_GA_LARGE, NACT) == P
    assert [spec[n][0] for n in ("conv1/w", "conv2/w", "conv3/w", "fc/w", "out/w", "out/b")] == [L.c1w, L.c2w, L.c3w, L.fcw, L.ow, L.ob]


def test_forward_every_layer_bit_exact(large_engine, oracle, big_noise):
    from dne_hip import _lib, ga_gpu
    e, O = large_engine, oracle
    L = O.layout(O.KIND_GA_LARGE, NACT)
    sb = ga_gpu.model_scale_by(NACT, _lib.KIND_GA_LARGE)
    parents = [(1234,), (3_000_000, (77, 0.004))]
    for slot, g in enumerate(parents, start=1):
        assert np.array_equal(e.ga_rebuild_powers(slot, g), O.ga_gpu_rebuild(big_noise, g, sb)), g
    # members: children of the two parents (one mutation applied on the fly) and an unmutated parent
    slots = np.array([1, 2, 1, 2, 1], np.int32)
    offs = np.array([555, 4_000_001, 2_222_222, 17, 0], np.int64)
    scales = np.array([0.002, 0.004, 0.0005, 0.003, 0.0], np.float32)
    e.set_members(slots, offs, scales)
    e.env_reset(np.array([3, 4, 5, 6, 7], np.uint32))
    for _ in range(3):                                     # a few steps so that the observations differ
        acts = e.act(5)[0]
        e.env_step(acts)
    obs = e.env_observation(5)
    acts, logits = e.act(5)
    for i in range(5):
        th = O.ga_gpu_rebuild(big_noise, parents[slots[i] - 1], sb)
        if scales[i] != 0:
            th = (th + scales[i] * big_noise[offs[i]:offs[i] + L.P]).astype(np.float32)   # base.py:141-142: theta + power * noise
        o1, o2, o3, o4, ol = O.forward_large_debug(L, th, obs[i])
        y1, y2, y3, y4 = e.debug_activations_large(i)
        assert np.array_equal(y1, o1), i
        assert np.array_equal(y2, o2), i
        assert np.array_equal(y3, o3), i
        assert np.array_equal(y4, o4), i
        assert np.array_equal(logits[i], ol) and acts[i] == int(np.argmax(ol)), i


@pytest.mark.parametrize("materialize,knobs", [("1", {}), ("0", {}),
                                               ("1", {"DNE_LFC_COLS_MAX": "0"}),                       # the streamed fc (k_lfc, one workgroup per CU by register footprint: the default above 96 members) at every count
                                               ("1", {"DNE_LFC_COLS_MAX": "0", "DNE_LFC_PAD": "0"}),   # ... its unpadded form
                                               ("1", {"DNE_LFC_COLS_MAX": "0", "DNE_LFC_PAD": "1"}),   # ... at most two per CU
                                               ("0", {"DNE_LFC_COLS_MAX": "0"})])                      # ... parent + noise rows on the fly
def test_genomes_evaluated_bit_exact(materialize, knobs, oracle, big_noise, monkeypatch):
    """materialize = 1 (default): children written out once per generation, the fc streams plain rows; 0: parent + noise rows on the fly"""
    from dne_hip import _lib, ga_gpu
    monkeypatch.setenv("DNE_GA_MATERIALIZE", materialize)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    e = _lib.Engine(_lib.KIND_GA_LARGE, NACT, max_members=16, record_bc=True)
    e.noise_upload(big_noise)
    e.ga_set_init_scale(ga_gpu.model_scale_by(NACT, _lib.KIND_GA_LARGE))
    O = oracle
    L = O.layout(O.KIND_GA_LARGE, NACT)
    sb = ga_gpu.model_scale_by(NACT, _lib.KIND_GA_LARGE)
    genomes = [(1234,), (200_000, (7, 0.002)), (2_900_000, (5, 0.004), (123_456, 0.001)), (200_000, (7, 0.002), (31_337, 0.003)),
               (4_500_000,), (200_000, (7, 0.002), (31_338, 0.003))]
    seeds = np.array([21, 22, 23, 24, 25, 26], np.uint32)
    ret, sg, ln, bc = e.ga_eval_powers(genomes, 45, seeds, want_bc=True)
    for i, g in enumerate(genomes):
        r, s, l, obc = O.rollout(L, O.ga_gpu_rebuild(big_noise, g, sb), None, seeds[i], 45, want_bc=True)
        assert (ret[i], sg[i], ln[i]) == (r, s, l) and np.array_equal(bc[i], obc), i
    assert len(set(ln.tolist())) > 1 or ln.max() == 45
    # next generation: children of cached parents, evaluated in another order
    kids = [genomes[1] + ((555, 0.0025),), genomes[3] + ((777_777, 0.0015),), genomes[1] + ((556, 0.0025),), genomes[4] + ((4_400_000, 0.002),)]
    ret, sg, ln = e.ga_eval_powers(kids, 45, seeds[:4])
    for i, g in enumerate(kids):
        assert (ret[i], sg[i], ln[i]) == O.rollout(L, O.ga_gpu_rebuild(big_noise, g, sb), None, seeds[i], 45)[:3], i
    with pytest.raises(_lib.DneError):
        e.ga_eval([[5, 6]], 0.002, 10, seeds[:1])            # es_distributed genomes (normc root) are GAAtariPolicy's
    e.close()


def test_deep_ga_driver_with_large_model(oracle, big_noise, tmp_path):
    """ga_gpu.main with exp['model'] = 'LargeModel' (configurations/ga_atari_config.json) on a tiny population: it runs, resumes
    from snapshot.pkl, and the elite it reports scores what the oracle scores for the same genome and environment seeds"""
    from dne_hip import _lib, es, ga_gpu
    e = _lib.Engine(_lib.KIND_GA_LARGE, NACT, max_members=12)
    try:
        noise = es.SharedNoiseTable.__new__(es.SharedNoiseTable)
        noise.noise = big_noise
        noise._engines = []
        exp = {"game": "frostbite", "model": "LargeModel", "num_validation_episodes": 2, "num_test_episodes": 2, "population_size": 10,
               "episode_cutoff_mode": 40, "timesteps": 1.5e9, "validation_threshold": 3, "mutation_power": 0.002, "selection_threshold": 3}
        test, val, state = ga_gpu.main(str(tmp_path), engine=e, noise=noise, seed=1, max_iters=2, **exp)
        assert state.it == 2 and np.isfinite(test) and len(state.population) == 10 and state.elite is not None
        assert all(len(o.seeds) >= 1 for o in state.population) and e.P == 4052658
        test2, val2, state2 = ga_gpu.main(str(tmp_path), engine=e, noise=noise, seed=1, max_iters=1, **exp)      # resumes at iteration 2
        assert state2.it == 3
    finally:
        e.close()


def test_second_larger_run_on_one_engine_keeps_parents_intact(oracle, big_noise):
    """Two runs on ONE engine, the second with more parents than the first reserved slots for: dne_ga_set_init_scale (every
    ga_gpu.main) frees all base slots, including the ones set aside for materialised children -- those must not stay reserved for
    children AND be handed to parents, or k_materialize_children writes a child over a live parent vector."""
    from dne_hip import _lib, ga_gpu
    e = _lib.Engine(_lib.KIND_GA_LARGE, NACT, max_members=16)
    try:
        e.noise_upload(big_noise)
        O = oracle
        L = O.layout(O.KIND_GA_LARGE, NACT)
        sb = ga_gpu.model_scale_by(NACT, _lib.KIND_GA_LARGE)
        e.ga_set_init_scale(sb)
        seeds = np.arange(16, dtype=np.uint32) + 40
        first = [(100_000 + 1000 * (i % 2), (7 + i, 0.002)) for i in range(6)]                # 2 parents, 6 materialised children
        e.ga_eval_powers(first, 20, seeds[:6])
        e.ga_set_init_scale(sb)                                                                 # what a second ga_gpu.main does
        second = [(200_000 + 50_000 * (i % 7), (900 + i, 0.003)) for i in range(14)]          # 7 parents, 14 children
        ret, sg, ln = e.ga_eval_powers(second, 30, seeds[:14])
        for i, g in enumerate(second):
            assert (ret[i], sg[i], ln[i]) == O.rollout(L, O.ga_gpu_rebuild(big_noise, g, sb), None, seeds[i], 30)[:3], i
        with pytest.raises(_lib.DneError, match="noise index"):                                # a corrupt seed fails cleanly, before any kernel
            e.ga_eval_powers([(100, (8_999_999, 0.002))], 5, seeds[:1])
        assert e.check_redzones() == 0
    finally:
        e.close()
