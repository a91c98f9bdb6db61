"""GPU: edge cases -- smallest and ragged inputs, limits, ties, refusals (the reference has no tests; these follow the
runtime asserts it does carry: es.py:196,246-248,297; ga.py:148-149)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NACT, NREF = 18, 16


@pytest.fixture(scope="module")
def eng(small_noise, oracle):
    from dne_hip import _lib
    es = _lib.Engine(_lib.KIND_ES, NACT, max_members=9, ref_count=NREF, record_bc=True, bc_max_steps=12)
    ga = _lib.Engine(_lib.KIND_GA, NACT, max_members=9)
    for e in (es, ga):
        e.noise_upload(small_noise)
    ref = oracle.get_ref_batch(seed=0, batch_size=NREF, nact=NACT)
    es.set_ref_batch(ref)
    es.set_theta(oracle.es_init_theta(oracle.layout(0, NACT), 0))
    yield es, ga, ref
    es.close(); ga.close()


def test_single_pair_and_single_step(eng, oracle, small_noise):
    es, _, ref = eng
    L = oracle.layout(0, NACT)
    th = es.get_theta()
    idx = np.array([3_000_000 - 1 - 58], np.int64)          # close to the end of the 4M-entry table
    idx[0] = small_noise.size - L.P                           # last legal index (es.py:66-67: randint(0, len - dim + 1))
    seeds = np.array([4294967295, 0], np.uint32)              # extreme seeds (noop counts 1 + seed % 30)
    for tsl in (1, 2, 12):
        ret, sg, ln = es.es_eval(idx, 0.02, tsl, seeds)
        oret, osg, oln = oracle.es_eval(L, th, small_noise, idx, 0.02, tsl, ref, seeds)
        assert np.array_equal(ret, oret) and np.array_equal(ln, oln) and np.array_equal(sg, osg)
        assert ln.max() <= tsl
    # odd member counts through the generic API (groups of one)
    for n in (1, 3):
        es.set_members(np.zeros(n, np.int32), np.full(n, 1234, np.int64), np.linspace(-0.02, 0.02, n).astype(np.float32))
        r, s, l = es.eval_members(n, 5, np.arange(n, dtype=np.uint32))
        assert r.shape == (n,) and l.max() <= 5


def test_refusals(eng, small_noise):
    from dne_hip import _lib
    es, ga, _ = eng
    P = es.P
    with pytest.raises(_lib.DneError):                         # more pairs than max_members allows
        es.es_eval(np.zeros(5, np.int64), 0.02, 3, np.zeros(10, np.uint32))
    with pytest.raises(_lib.DneError):                         # one past the last legal noise index
        es.es_eval(np.array([small_noise.size - P + 1], np.int64), 0.02, 3, np.zeros(2, np.uint32))
    with pytest.raises(_lib.DneError):
        es.es_eval(np.array([-1], np.int64), 0.02, 3, np.zeros(2, np.uint32))
    with pytest.raises(_lib.DneError):                         # non-positive timestep limit
        es.es_eval(np.array([0], np.int64), 0.02, 0, np.zeros(2, np.uint32))
    with pytest.raises(_lib.DneError):                         # GA entry point on an ES engine and vice versa
        es.ga_eval([[1]], 0.005, 3, np.zeros(1, np.uint32))
    with pytest.raises(_lib.DneError):
        ga.es_eval(np.array([0], np.int64), 0.02, 3, np.zeros(2, np.uint32))
    with pytest.raises(_lib.DneError):
        ga.set_ref_batch(np.zeros((NREF, 84, 84, 4), np.uint8))
    with pytest.raises(_lib.DneError):                         # wrong reference-batch size
        es.set_ref_batch(np.zeros((NREF + 16, 84, 84, 4), np.uint8))
    with pytest.raises(_lib.DneError):
        es.centered_ranks(np.zeros(1, np.float32))            # es.py:83 divides by size - 1
    with pytest.raises(_lib.DneError):
        es.ga_select(np.zeros(3, np.float32), 4)
    with pytest.raises(_lib.DneError):                         # BCs requested from an engine without record_bc
        ga2 = None
        from dne_hip import _lib as L2
        e2 = L2.Engine(L2.KIND_ES, NACT, max_members=2, ref_count=NREF)
        try:
            e2.noise_upload(small_noise); e2.set_ref_batch(np.zeros((NREF, 84, 84, 4), np.uint8))
            e2.es_eval(np.array([0], np.int64), 0.02, 3, np.zeros(2, np.uint32), want_bc=True)
        finally:
            e2.close()


def test_reduce_edges(eng, oracle, small_noise):
    es, _, _ = eng
    P = es.P
    assert np.array_equal(es.centered_ranks(np.array([5.0, 5.0], np.float32)), np.array([-0.5, 0.5], np.float32))
    x = np.array([[np.inf, -np.inf], [0.0, -0.0], [1e-45, 3.4e38]], np.float32)   # infinities, signed zero, denormal, max
    assert np.array_equal(es.centered_ranks(x), oracle.centered_ranks(x.reshape(-1)).reshape(x.shape))
    g = es.weighted_sum(np.array([7], np.int64), np.array([2.5], np.float32), 1.0)
    assert np.array_equal(g, np.float32(2.5) * small_noise[7:7 + P])
    g0 = es.weighted_sum(np.array([7, 9], np.int64), np.zeros(2, np.float32), 4.0)
    assert not g0.any()
    scores = np.array([10, 30, 30, 10, 30], np.float32)       # ties at and above the cut
    assert es.ga_select(scores, 1).tolist() == [1] and es.ga_select(scores, 4).tolist() == [1, 2, 4, 0]
    assert es.ga_select(scores, 5).tolist() == [1, 2, 4, 0, 3]
    assert es.ga_select(np.array([3.0], np.float32), 1).tolist() == [0]


def test_ga_ragged_chains_and_elite_reuse(eng, oracle, small_noise):
    _, ga, _ = eng
    L = oracle.layout(1, NACT)
    hi = small_noise.size - L.P
    chains = [[5], [5, 77], [5, 77, 1999], [hi], [hi, 0, hi, 0, hi, 0, hi], [123456, 5]]   # lengths 1..7, shared prefixes
    seeds = np.arange(len(chains), dtype=np.uint32) * 7919
    ret, sg, ln = ga.ga_eval(chains, 0.005, 9, seeds)
    for i, c in enumerate(chains):
        th = oracle.ga_rebuild(L, small_noise, c, 0.005)
        assert (ret[i], sg[i], ln[i]) == oracle.rollout(L, th, None, seeds[i], 9)[:3], c
    # same chains again: everything is served from the parent cache, results identical
    ret2, sg2, ln2 = ga.ga_eval(chains, 0.005, 9, seeds)
    assert np.array_equal(ret, ret2) and np.array_equal(ln, ln2)
    with pytest.raises(Exception):
        ga.ga_eval([[]], 0.005, 3, np.zeros(1, np.uint32))


def test_novelty_length_cases(eng, oracle):
    es, _, _ = eng
    rs = np.random.RandomState(0)
    bc = rs.randint(0, 256, (9, 128)).astype(np.uint8)
    arch = [rs.randint(0, 256, (n, 128)).astype(np.uint8) for n in (1, 9, 8, 10, 300)]   # shorter, equal, longer (nses.py:12-20)
    arch.append(bc.copy())                                                                  # identical trajectory: distance 0
    for k in (1, 3, 6, 50):                                                                 # k larger than the archive too
        assert es.novelty(arch, bc, k) == oracle.novelty(arch, bc, k)
    assert es.novelty([bc.copy()], bc, 1) == 0.0
    assert es.novelty(arch, bc[:1], 2) == oracle.novelty(arch, bc[:1], 2)                   # single-row BC


def test_device_trajectories_after_longer_run(eng, oracle, small_noise):
    """trajectories kept on the device (no download, no clearing): rows past an episode's length hold the previous
    run's data and must not influence novelty; bytes past the emulator's live RAM stay zero"""
    es, _, ref = eng
    L = oracle.layout(0, NACT)
    th = es.get_theta()
    idx = np.array([100, 2000, 30000], np.int64)
    seeds = np.arange(6, dtype=np.uint32) + 11
    _, _, ln_long, bc_long = es.es_eval(idx, 0.02, 12, seeds, want_bc=True)        # fills all 12 rows
    assert ln_long.max() == 12 and bc_long.reshape(6, 12, 128)[:, :, 40:].max() == 0
    ret, sg, ln = es.es_eval(idx[::-1].copy(), 0.02, 5, seeds[::-1].copy())        # shorter, different members, stays on the device
    rs = np.random.RandomState(1)
    arch = [rs.randint(0, 256, (n, 128)).astype(np.uint8) for n in (3, 5, 12)]
    nov = es.novelty_batch(arch, ln.reshape(-1), 2)
    for i in range(3):
        for s in range(2):
            thp = oracle.perturb(th, small_noise, idx[::-1][i], 0.02, 1 if s == 0 else -1)
            r, _, l, bc = oracle.rollout(L, thp, ref, seeds[::-1][2 * i + s], 5, want_bc=True)
            assert l == ln[i, s] and nov[2 * i + s] == oracle.novelty(arch, bc, 2)


# Settings that put a kernel or schedule NO default path launches onto the test population (the A/B library of DESIGN.md section 4):
# they run with `-m "gpu and variants"` (or DNE_TEST_VARIANTS=1) only; everything else forces a DEFAULT-path kernel onto a population
# small enough for the oracle and stays in the default GPU suite.
_VARIANT_KEYS = {"DNE_FC2_MIN", "DNE_DUO_LAG", "DNE_DUO_HEAD_FUSED", "DNE_DUO_SWEEP", "DNE_DUO_SYNC", "DNE_FC_DUO",
                 "DNE_FC_PAIRS", "DNE_FC_RB", "DNE_HEAD_THREADS", "DNE_TAIL_TABLE", "DNE_SPEC_CONV1", "DNE_SPEC_BANDS", "DNE_TAIL_FUSED_MAX",
                 "DNE_CONV1_FPW", "DNE_CONV1_SHARED", "DNE_BAND_THREADS", "DNE_FC_SUB_SPW", "DNE_FC_SUB_NSUB", "DNE_DUO_FAT", "DNE_GA_MATERIALIZE", "DNE_SUB_RENDER_FUSED"}


def _knob_params(knob_list):
    return [pytest.param(k, marks=pytest.mark.variants) if _VARIANT_KEYS & set(k) else k for k in knob_list]


_ES_STEP_KNOBS = [
    {"DNE_FC2_MIN": "2", "DNE_FC_TAIL_MAX": "1"},                       # k_fc2 (two pairs per work item; odd count: repeated pair)
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1"},                    # table-ordered units: k_unit_order + k_fc_duo (one unit per wave below 1500 pairs) + k_out
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SOLO_BELOW": "0", "DNE_DUO_LAG": "3"},   # ... two units per wave, the second trailing further (a schedule, not arithmetic)
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_NSUB": "2", "DNE_DUO_SOLO_BELOW": "0"},   # ... two windows, each with its own unit order
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SOLO_BELOW": "0"},   # ... always two units per wave (the test population is below the default switch to one)
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_HEAD_FUSED": "1"},   # ... the fused head + emulator launch behind it instead of k_out + k_env_logic
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SOLO_BELOW": "0", "DNE_DUO_SWEEP": "0"},   # ... without the workgroup's common table timeline (round 2's schedule: every wave starts its duo at once), two units per wave
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SWEEP": "0"},   # ... one unit per wave
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SOLO_BELOW": "0", "DNE_DUO_SWEEP": "1"},   # ... the timeline only for two units per wave (default 2: also for one)
    {"DNE_FC_SUB": "2", "DNE_FC_SUB_MIN": "2"},                         # the sub-slice fc (k_fc_sub: one wave per 128 / 120-row chain, folded by k_out<.., SUB>) for pairs, two windows
    {"DNE_FC_SUB": "2", "DNE_FC_SUB_MIN": "2", "DNE_FC_SUB_NSUB": "1", "DNE_FC_SUB_SPW": "2"},   # ... one window, two chains per wave
    {"DNE_FC_SUB": "2", "DNE_FC_SUB_MIN": "2", "DNE_FC_SUB_SPW": "8"},   # ... a whole quarter per wave
    {"DNE_FC_SUB": "2", "DNE_FC_SUB_MIN": "2", "DNE_SUB_RENDER_FUSED": "1"},   # ... head + emulator + renderer in one launch behind it (three launches per window and lock-step; round 6, measured slower: not the default)
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SOLO_BELOW": "0", "DNE_DUO_FAT": "1"},   # ... the form that the hardware places once per CU (register footprint past 256), two units per wave
    {"DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_FAT": "1", "DNE_NSUB": "2"},   # ... one unit per wave, two windows
    {"DNE_FC_RING": "1", "DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SOLO_BELOW": "0", "DNE_RING_MIN": "0"},   # k_fc_ring (round 5) through its DEFAULT gate (dense enough, enough pairs): the workgroup's noise rows through an LDS ring (LDS-DMA), base rows from the column-permuted copy; one unit per wave, eight per workgroup
    {"DNE_FC_RING": "2", "DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1"},   # ... in the sparse regime too (k_y2_activate behind the tail's convolution kernels)
    {"DNE_FC_RING": "2", "DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_CONV_FUSED_MIN": "1"},   # ... behind k_conv12, which leaves relu(bn2(y2)) itself
    {"DNE_FC_RING": "2", "DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_CONV_FUSED": "0", "DNE_CONV12T_MAX": "0"},   # ... behind k_conv1 + k_conv2
    {"DNE_FC_RING": "1", "DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_DUO_SOLO_BELOW": "0", "DNE_RING_MIN": "0", "DNE_NSUB": "2"},   # ... two windows, each with its own unit order
    {"DNE_FC_RING": "2", "DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1", "DNE_RING_PRE": "0"},   # ... scaling its noise rows itself (k_fc_ring<false>: the path an engine takes when the table's scaled copy does not fit; the cases above stream the copy)
    {"DNE_BURST": "5", "DNE_BURST_TAIL": "40"},                         # compaction of the active list every 5 / 40 lock-steps instead of 16
    {"DNE_FC_DUO": "0", "DNE_FC2_MIN": "2", "DNE_FC_TAIL_MAX": "1"},    # the duo path switched off: k_fc2
    {"DNE_FC_PAIRS": "1", "DNE_FC_TAIL_MAX": "1"},                      # k_fc<2> streaming kernel
    {"DNE_FC_PAIRS": "1", "DNE_FC_TAIL_MAX": "1", "DNE_FC_RB": "2"},    # ... with 2-row batches
    {"DNE_SPEC_MAX": "0"},                                              # no speculative tail: k_tail_step + banded render (the default below steps every action under the forward pass)
    {"DNE_SPEC_MAX": "0", "DNE_TAIL_TABLE": "0"},                       # ... member descriptors read from memory instead of the kernel arguments
    {"DNE_SPEC_MAX": "0", "DNE_HEAD_THREADS": "256"},                   # ... the policy head without its fifth wave: one lane steps the emulator after the argmax
    {"DNE_TAIL_TABLE": "0"},                                            # speculative tail without the kernel-argument table
    {"DNE_SPEC_CONV1": "0"},                                            # speculative tail without the candidate conv1 (every lock-step starts at conv1)
    {"DNE_SPEC_MAX": "4"},                                              # speculative only for the last two pairs (default: the last four)
    {"DNE_SPEC_MAX": "64", "DNE_SPEC_BANDS": "2"},                      # speculative from the first lock-step on, two render workgroups per candidate
    {"DNE_SPEC_MAX": "0", "DNE_TAIL_FUSED_MAX": "0"},                   # k_fc_tail + k_out + separate emulator / render launches
    {"DNE_SPEC_MAX": "0", "DNE_FC_QUAD_MAX": "0"},                      # k_fc_tail down to the last pair
    {"DNE_SPEC_MAX": "0", "DNE_FC_QUAD_MAX": "0", "DNE_TAIL_TABLE": "0"},   # ... reading the member descriptors from memory
    {"DNE_SPEC_MAX": "0", "DNE_FC_QUAD_MAX": "64"},                     # k_fc_quad (64 workgroups per pair) at every count
    {"DNE_SPEC_MAX": "0", "DNE_FC_QUAD_MAX": "0", "DNE_FC_TAILK_MAX": "0"},   # k_fc_cols (4 workgroups per pair, the form for > 32 pairs per window) at every count
    {"DNE_SPEC_MAX": "8"},                                              # speculative tail from four pairs on (round 2's default; now two)
    {"DNE_RENDER_BANDS": "1"},                                          # k_fc_tail + tail step rendering in place
    {"DNE_SPEC_MAX": "0", "DNE_CONV12T_MAX": "0"},                      # the tail's convolutions as two launches (k_conv1 over 7, k_conv2 over 4 workgroups per member) instead of k_conv12t
    {"DNE_SPEC_MAX": "0", "DNE_CONV12T_MAX": "0", "DNE_TAIL_TABLE": "0"},
    {"DNE_SPEC_MAX": "0", "DNE_CONV12T_MAX": "100000", "DNE_CONV_FUSED": "0"},   # k_conv12t at every count
    {"DNE_CONV1_FPW": "1"},                                             # reference-pass conv1 with one frame per workgroup (default 8)
    {"DNE_CONV1_FPW": "4"},
    {"DNE_CONV1_SHARED": "0"},                                          # reference-pass conv1 per member (k_conv1_ref<8>) instead of the shared float image
    {"DNE_CONV_SPLIT_MAX": "0"},                                        # convolutions with 4 / 2 workgroups per member instead of 7 / 4
    {"DNE_RENDER_BANDS": "7", "DNE_BAND_THREADS": "1024"},              # frame split over 7 workgroups
    {"DNE_NSUB": "3", "DNE_FC_TAIL_MAX": "2", "DNE_FC2_MIN": "4"},      # three windows, k_fc2 / k_fc / tail kernels as the list shrinks
    {"DNE_RENDER_BANDS": "2"},                                          # two render workgroups per member
    {"DNE_CONV_FUSED_MIN": "1"},                                        # conv1 + conv2 in one kernel (k_conv12) at every count
    {"DNE_CONV_FUSED": "0", "DNE_CONV_SPLIT_MAX": "0"},                 # never: separate k_conv1 / k_conv2 launches
]


@pytest.mark.parametrize("knobs", _knob_params(_ES_STEP_KNOBS))
def test_every_step_kernel_variant_is_bit_exact(knobs, oracle, small_noise, monkeypatch):
    """the engine picks its lock-step kernels by active count; the tuning knobs force each variant onto a population small
    enough for the oracle (5 pairs = odd count, episodes of different lengths so that the active list shrinks)"""
    from dne_hip import _lib
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    e = _lib.Engine(_lib.KIND_ES, NACT, max_members=10, ref_count=NREF)
    try:
        e.noise_upload(small_noise)
        ref = oracle.get_ref_batch(seed=0, batch_size=NREF, nact=NACT)
        e.set_ref_batch(ref)
        L = oracle.layout(0, NACT)
        th = oracle.es_init_theta(L, 0)
        e.set_theta(th)
        idx = np.array([11, 222_222, 2_900_001, 1_234_567, 42], np.int64)
        seeds = (np.arange(10, dtype=np.uint32) * 2654435761).astype(np.uint32)
        ret, sg, ln = e.es_eval(idx, 0.02, 150, seeds)
        oret, osg, oln = oracle.es_eval(L, th, small_noise, idx, 0.02, 150, ref, seeds)
        assert np.array_equal(ln, oln) and np.array_equal(ret, oret) and np.array_equal(sg, osg), knobs
        assert len(set(ln.reshape(-1).tolist())) > 3          # the list did shrink step by step
    finally:
        e.close()


def test_ring_scaled_table_follows_sigma_and_table_writes(oracle, small_noise, monkeypatch):
    """k_fc_ring<true> streams a copy of the noise table scaled by the evaluation's sigma (DNE_RING_PRE): the copy is
    made again when the next evaluation comes with another sigma (a scheduled mutation power: dne_hip/es_gpu.py), with sigma 0, and after
    a part of the table is rewritten (dne_noise_write) -- every evaluation against the oracle."""
    from dne_hip import _lib
    for k, v in {"DNE_FC_RING": "2", "DNE_FC_DUO_MIN": "2", "DNE_FC_TAIL_MAX": "1"}.items():
        monkeypatch.setenv(k, v)
    e = _lib.Engine(_lib.KIND_ES, NACT, max_members=10, ref_count=NREF)
    try:
        noise = small_noise.copy()
        e.noise_upload(noise)
        ref = oracle.get_ref_batch(seed=0, batch_size=NREF, nact=NACT)
        e.set_ref_batch(ref)
        L = oracle.layout(0, NACT)
        th = oracle.es_init_theta(L, 0)
        e.set_theta(th)
        idx = np.array([11, 222_222, 2_900_001, 1_234_567, 42], np.int64)
        seeds = (np.arange(10, dtype=np.uint32) * 2654435761).astype(np.uint32)

        def check(sigma, steps):
            ret, sg, ln = e.es_eval(idx, sigma, steps, seeds)
            oret, osg, oln = oracle.es_eval(L, th, noise, idx, sigma, steps, ref, seeds)
            assert np.array_equal(ln, oln) and np.array_equal(ret, oret) and np.array_equal(sg, osg), sigma
        check(0.02, 40)
        check(0.05, 40)                    # another sigma on the same engine: the scaled copy is rebuilt
        check(0.0, 25)                     # sigma 0: every member is theta itself
        check(0.02, 40)
        chunk = np.random.RandomState(7).randn(300_000).astype(np.float32)
        noise[200_000:500_000] = chunk     # rows of the second pair's slice change under the copy
        e.noise_write(200_000, chunk)
        check(0.02, 40)
    finally:
        e.close()


def _crafted_rams(rs, n):
    """valid SynthAtari RAM snapshots that visit every branch of the renderer: sprite on every row
    band and wrapping over the screen edge, blink on/off, temperature / lives / score HUD, igloo 0..16 with and without the
    door, both sky colours, hazards active or not at any x, floes visited or not at any offset"""
    ram = np.zeros((n, 128), np.uint8)
    ram[:, 0:2] = rs.randint(0, 256, (n, 2))                      # frame counter (bit 2 drives the blink)
    ram[:, 2:6] = rs.randint(0, 256, (n, 4))                      # LCG state (not rendered)
    prow = rs.randint(0, 5, n); ram[:, 7] = prow
    ram[:, 6] = np.where(prow == 0, rs.randint(8, 145, n), rs.randint(0, 160, n))
    ram[:, 8] = rs.randint(0, 4, n)                               # lives
    ram[:, 10] = rs.randint(0, 46, n)                             # temperature
    ram[:, 11] = rs.randint(0, 13, n)
    ram[:, 12:16] = rs.randint(0, 160, (n, 4))                    # floe offsets
    ram[:, 16:20] = rs.randint(0, 2, (n, 4))                      # visited
    ram[:, 20] = rs.choice([0, 1, 3, 4, 7, 8, 11, 12, 15, 16], n) # igloo blocks
    ram[:, 21] = rs.randint(0, 6, n)                              # level (sky colour, speed)
    ram[:, 22:25] = rs.randint(0, 256, (n, 3))                    # score / 10
    ram[:, 25] = rs.choice([0, 0, 1, 64, 128], n)                 # freeze counter
    ram[:, 26:30] = rs.randint(0, 160, (n, 4))                    # hazard x
    ram[:, 30:34] = rs.randint(0, 2, (n, 4))                      # hazard active
    ram[:, 34:38] = rs.randint(0, 2, (n, 4))
    ram[:, 38] = rs.randint(0, 18, n); ram[:, 39] = rs.randint(0, 48, n)
    return ram


def test_renderer_on_crafted_states(eng, oracle):
    """observation = warp(max(frame(ram_prev), frame(ram_cur))) for injected RAM pairs vs the oracle's renderer + PIL restatement"""
    es, _, _ = eng
    rs = np.random.RandomState(7)
    pal = oracle.palette()
    for _ in range(12):
        n = 9
        prev = _crafted_rams(rs, n)
        cur = prev.copy()
        for i in range(n):                                        # half of them: a genuine next frame; the rest: an unrelated state
            if i % 2 == 0:
                oracle.raw_frame(cur[i], int(rs.randint(0, 18)))
            else:
                cur[i] = _crafted_rams(rs, 1)[0]
        es.env_set_ram(prev, cur)
        obs = es.env_observation(n)
        assert np.array_equal(es.env_ram(n), cur)
        for i in range(n):
            rgb = np.maximum(pal[oracle.raw_render(prev[i])], pal[oracle.raw_render(cur[i])])
            want = oracle.warp_rgb(rgb)
            assert np.array_equal(obs[i, :, :, 0], want), (i, prev[i, :40].tolist(), cur[i, :40].tolist())
            assert all(np.array_equal(obs[i, :, :, c], want) for c in (1, 2, 3))


def test_emulator_step_from_crafted_states(eng, oracle):
    """one wrapped step (4 raw frames, early stop on game over) from injected states, every action: RAM, reward and the
    game-over flag vs the oracle's emulator -- reaches the branches random play rarely does (level completion, last life,
    temperature death, direction reversal, hazard collisions on every row)"""
    es, _, _ = eng
    rs = np.random.RandomState(11)
    n = 9
    seen = {"level_up": 0, "game_over": 0, "direction_flip": 0, "reward": 0}
    for rep in range(40):
        cur = _crafted_rams(rs, n)
        if rep % 4 == 0:                                          # stage the rare events on purpose
            cur[:, 20] = 16; cur[:, 7] = 0; cur[:, 6] = rs.randint(104, 145, n); cur[:, 25] = 0; cur[:, 11] = 0   # finished igloo, at the door
        if rep % 4 == 1:
            cur[:, 8] = 0; cur[:, 10] = 1; cur[:, 39] = 47                                                       # last life, about to freeze
        if rep % 4 == 2:
            cur[:, 25] = 0; cur[:, 11] = 0; cur[:, 7] = rs.randint(1, 5, n); cur[:, 20] = rs.randint(1, 17, n)   # on the ice, may FIRE
        cur[:, 9] = 0
        actions = rs.randint(0, 18, n).astype(np.int32)
        if rep % 4 == 0:
            actions[:] = rs.choice([2, 6, 7, 10, 14, 15], n)      # UP-ish
        es.env_reset(np.arange(n, dtype=np.uint32))               # clears done / length bookkeeping
        es.env_set_ram(cur, cur)
        rew, done = es.env_step(actions)
        got = es.env_ram(n)
        for i in range(n):
            ram = cur[i].copy()
            tot, over = 0, False
            for _ in range(4):                                    # atari_wrappers.py:95-107
                tot += oracle.raw_frame(ram, int(actions[i]))
                if ram[9]:
                    over = True
                    break
            assert np.array_equal(got[i], ram), (rep, i, int(actions[i]), cur[i, :40].tolist())
            assert rew[i] == tot and bool(done[i]) == over, (rep, i, rew[i], tot, done[i], over)
            seen["level_up"] += int(ram[21] > cur[i, 21]); seen["game_over"] += int(over); seen["reward"] += int(tot > 0)
            seen["direction_flip"] += int(np.any(ram[34:38] != cur[i, 34:38]))
    assert all(v > 0 for v in seen.values()), seen


@pytest.mark.parametrize("nref,members", [(8, 3), (24, 9), (40, 5), (32, 17), (64, 2)])
def test_reference_batch_sizes(nref, members, oracle, small_noise):
    """Virtual batch norm over reference batches other than the reference's 128 frames (es.py:160-162 takes any batch_size): multiples
    of 8 that are / are not multiples of 16 (the shared-image conv1 takes 16 or 8 frames per workgroup), the matrix-core fc path
    (16, 32, 64, 128 frames) and the generic one, member counts that do not fill the last eight-member workgroup."""
    from dne_hip import _lib
    e = _lib.Engine(_lib.KIND_ES, NACT, max_members=32, ref_count=nref)
    try:
        e.noise_upload(small_noise)
        L = oracle.layout(0, NACT)
        th = oracle.es_init_theta(L, 0)
        e.set_theta(th)
        ref = oracle.get_ref_batch(seed=3, batch_size=nref, nact=NACT)
        e.set_ref_batch(ref)
        rs = np.random.RandomState(nref)
        off = rs.randint(0, small_noise.size - L.P, members).astype(np.int64)
        scale = rs.choice([0.02, -0.02, 0.0, 0.1], members).astype(np.float32)
        e.set_members(np.zeros(members, np.int32), off, scale)
        e.ref_pass(members)
        bn, mom = e.get_bn(members), e.get_bn_moments(members)
        for i in (0, members - 1, members // 2):
            obn, omom = oracle.es_ref_pass_moments(L, th + np.float32(scale[i]) * small_noise[off[i]:off[i] + L.P], ref)
            assert np.array_equal(bn[i], obn) and np.array_equal(mom[i], omom), (nref, i)
    finally:
        e.close()


_GA_STEP_KNOBS = [
    {"DNE_SPEC_MAX": "0"},                                                                            # GA tail without speculation
    {"DNE_SPEC_MAX": "64"},                                                                           # ... speculative from the first lock-step
    {"DNE_GA_MATERIALIZE": "0"},                                                                      # children NOT written out: parent row + noise row on the fly
    {"DNE_GA_MATERIALIZE": "0", "DNE_FC_TAIL_MAX": "1"},                                              # ... through the streaming fc at every count
    {"DNE_FC_TAIL_MAX": "1"},                                                                         # children written out (default), noise-free streaming fc at every count
    {"DNE_FC_TAIL_MAX": "1", "DNE_FC_RB": "8"},                                                       # ... with 8-row batches
    {"DNE_SPEC_MAX": "0", "DNE_FC_QUAD_MAX": "0", "DNE_FC_TAILK_MAX": "0"},                           # k_fc_cols<1>
    {"DNE_SPEC_MAX": "0", "DNE_FC_QUAD_MAX": "0"},                                                    # k_fc_tail<1> (noise-free form: children written out)
    {"DNE_SPEC_MAX": "0", "DNE_FC_QUAD_MAX": "64"},                                                   # k_fc_quad<1>
    {"DNE_SPEC_MAX": "0", "DNE_CONV12T_MAX": "0"},                                                    # k_conv1 + k_conv2 in the tail instead of k_conv12t<false>
    {"DNE_FC_SUB_MIN": "2"},                                                                          # the mid range's sub-slice fc (k_fc_sub<1, false, false> on written-out children) at every count
    {"DNE_FC_SUB_MIN": "2", "DNE_FC_SUB_SPW": "4", "DNE_FC_SUB_NSUB": "3"},                           # ... four chains per wave, three windows
    {"DNE_FC_SUB": "0"},                                                                              # ... and switched off
]


@pytest.mark.parametrize("knobs", _knob_params(_GA_STEP_KNOBS))
def test_ga_step_kernel_variants_are_bit_exact(knobs, oracle, small_noise, monkeypatch):
    """the GA evaluation (single members, one base vector per parent, final-RAM behaviour characterisation) through the kernel
    variants the ES test above cannot reach"""
    from dne_hip import _lib
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    O = oracle
    e = _lib.Engine(_lib.KIND_GA, NACT, max_members=16, record_bc=True)
    try:
        e.noise_upload(small_noise)
        L = O.layout(O.KIND_GA, NACT)
        sigma, tslimit = 0.005, 70
        rs = np.random.RandomState(11)
        hi = small_noise.size - L.P + 1
        gen0 = [[int(rs.randint(hi))] for _ in range(7)]
        seeds = rs.randint(0, 2 ** 31, 7).astype(np.uint32)
        ret, sg, ln, bc = e.ga_eval(gen0, sigma, tslimit, seeds, want_bc=True)
        for i, chain in enumerate(gen0):
            r, s, l, obc = O.rollout(L, O.ga_rebuild(L, small_noise, chain, sigma), None, seeds[i], tslimit, want_bc=True)
            assert (ret[i], sg[i], ln[i]) == (r, s, l) and np.array_equal(bc[i], obc), (knobs, i)
        parents = [gen0[2], gen0[5]]
        gen1 = [parents[i % 2] + [int(rs.randint(hi))] for i in range(7)]
        seeds = rs.randint(0, 2 ** 31, 7).astype(np.uint32)
        ret, sg, ln = e.ga_eval(gen1, sigma, tslimit, seeds)
        for i, chain in enumerate(gen1):
            assert (ret[i], sg[i], ln[i]) == O.rollout(L, O.ga_rebuild(L, small_noise, chain, sigma), None, seeds[i], tslimit)[:3], (knobs, i)
    finally:
        e.close()
