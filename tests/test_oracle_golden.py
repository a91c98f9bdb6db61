"""CPU: pin the oracle against vectors produced by the REAL reference code
(tests/golden/make_golden.py -> reference_vectors.npz) and against PIL/numpy/torch run here."""
import hashlib
import os

import numpy as np
import pytest


def test_noise_stream_head(golden, small_noise):
    # SURVEY 8c golden bits for float32(RandomState(123).randn(...))
    bits = small_noise[:8].view(np.uint32)
    assert bits.tolist() == [3213555186, 1065308680, 1049682575, 3217083972, 3205766949, 1070817862,
                             3223015094, 3202062960]
    assert np.array_equal(golden["noise_small_head"], small_noise[:8])
    if "noise_full_head" in golden.files:
        assert np.array_equal(golden["noise_full_head"], small_noise[:8])
        assert golden["noise_full_tail"].view(np.uint32).tolist() == [3158718917, 3219242230, 1075127793]


def test_sample_index_golden(golden):
    # es.py:66-67 with the real table length; SURVEY 8c
    for dim in (1009058, 1008450):
        assert golden["sample_index_%d" % dim][:4].tolist() == [209652396, 130329135, 118924917, 136432832]


def test_centered_ranks(golden, oracle):
    for tag in ("small", "distinct"):
        x = golden["ranks_%s_in" % tag]
        y = oracle.centered_ranks(x.reshape(-1)).reshape(x.shape)
        assert y.dtype == np.float32
        if tag == "small":
            # ties resolved in flat-index order
            assert np.allclose(y, np.array([[-0.1, -0.5], [-0.3, 0.5], [0.1, 0.3]], np.float32), atol=1e-7)
            assert np.array_equal(np.sort(y.reshape(-1)), np.sort(golden["ranks_small_out"].reshape(-1)))
        else:
            assert np.array_equal(y, golden["ranks_%s_out" % tag])
    # tied returns: reference tie order is implementation-defined; the multiset of ranks
    # and every untied element must agree, tied groups must hold the same rank set
    x = golden["ranks_tied_in"].reshape(-1)
    ref = golden["ranks_tied_out"].reshape(-1)
    y = oracle.centered_ranks(x)
    assert np.array_equal(np.sort(y), np.sort(ref))
    for v in np.unique(x):
        m = x == v
        assert np.array_equal(np.sort(y[m]), np.sort(ref[m]))
    # stability: within a tie group ranks increase with flat index
    for v in np.unique(x):
        assert np.all(np.diff(y[x == v]) > 0)


def test_weighted_sum(golden, oracle):
    noise = np.random.RandomState(123).randn(2_000_000).astype(np.float32)
    P = int(golden["ws_P"])
    g = oracle.weighted_sum(noise, golden["ws_idx"], golden["ws_w"], P, 1.0)
    ref = golden["ws_g"]
    assert g.dtype == np.float32 and g.shape == (P,)
    # BLAS order vs i-ordered fmaf chain: tolerance class of the north star (1e-5 on the update vector)
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= 2e-6 * scale + 1e-6
    assert np.abs(g - golden["ws_g_b7"]).max() <= 2e-6 * scale + 1e-6
    # float64 ground truth is closer to the oracle than 1e-5 as well
    g64 = np.zeros(P)
    for w, i in zip(golden["ws_w"], golden["ws_idx"]):
        g64 += float(w) * noise[i:i + P].astype(np.float64)
    assert np.abs(g - g64).max() <= 2e-6 * scale + 1e-6


def test_optimizers(golden, oracle):
    theta0, gs = golden["opt_theta0"], golden["opt_gs"]
    for name, mk in (("adam", lambda: oracle.Adam(theta0, 0.01)), ("sgd", lambda: oracle.SGD(theta0, 0.01, 0.9))):
        opt = mk()
        for t, g in enumerate(gs):
            ratio, theta = opt.update(g, 0.005)
            ref = golden["opt_%s_thetas" % name][t]
            # the reference run under numpy>=2 computes Adam's step in float64; the oracle
            # forces float32 as numpy 1.12 did -> agreement to float32 rounding of theta
            assert np.abs(theta - ref).max() <= 1.5e-7 * max(1.0, np.abs(ref).max()) * 4
            assert abs(ratio - golden["opt_%s_ratios" % name][t]) <= 1e-5 * golden["opt_%s_ratios" % name][t]


def test_adam_float32_restatement(golden, oracle):
    # bit-exact against optimizers.py:45-50 restated with every scalar forced to float32
    theta0, gs = golden["opt_theta0"], golden["opt_gs"]
    th = theta0.copy(); m = np.zeros_like(th); v = np.zeros_like(th)
    opt = oracle.Adam(theta0, 0.01)
    f = np.float32
    for t, g in enumerate(gs, 1):
        gg = -g + f(0.005) * th
        a = 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        m = f(0.9) * m + f(1 - 0.9) * gg
        v = f(0.999) * v + f(1 - 0.999) * (gg * gg)
        step = f(-a) * m / (np.sqrt(v) + f(1e-8))
        th = th + step
        assert th.dtype == np.float32
        _, theta = opt.update(g, 0.005)
        assert np.array_equal(theta, th)


def test_novelty(golden, oracle):
    lens = golden["nov_lens"]
    arch, o = [], 0
    for n in lens:
        arch.append(golden["nov_arch"][o:o + n]); o += n
    bc = golden["nov_bc"]
    d = np.array([oracle.bc_distance(a, bc) for a in arch])
    assert np.allclose(d, golden["nov_dists"], rtol=1e-13, atol=0)
    assert np.isclose(oracle.novelty(arch, bc, 10), float(golden["nov_k10"]), rtol=1e-13)
    assert np.isclose(oracle.novelty(arch, bc, 3), float(golden["nov_k3"]), rtol=1e-13)


def test_normc(golden, oracle):
    import ctypes as C
    lib = oracle.lib()
    # tensor-level check through the GA layout: place tensors at the layout offsets
    L = oracle.layout(oracle.KIND_GA, 18)
    th = np.random.RandomState(9).randn(L.P).astype(np.float32)
    a_in, c_in = golden["normc_a_in"], golden["normc_c_in"]
    th[L.c1w:L.c1w + 4096] = a_in.reshape(-1)
    th[L.ow:L.ow + 256 * 18] = c_in.reshape(-1)
    out = oracle.ga_normc(L, th)
    assert np.array_equal(out[L.c1w:L.c1w + 4096], golden["normc_a_out"].reshape(-1))
    assert np.array_equal(out[L.ow:L.ow + 256 * 18], golden["normc_c_out"].reshape(-1))
    for off, n in ((L.c1b, 16), (L.c2b, 32), (L.fcb, 256), (L.ob, 18)):
        assert not out[off:off + n].any()
    # K = 3872 sequential column sums (fc): numpy restatement on the full fc tensor
    w = th[L.fcw:L.fcw + 3872 * 256].reshape(3872, 256).copy()
    w *= 1.0 / np.sqrt(np.square(w).sum(axis=0, keepdims=True))
    assert np.array_equal(out[L.fcw:L.fcw + 3872 * 256], w.reshape(-1))
    b_in = golden["normc_b_in"]
    o = b_in.copy(); o *= 1.0 / np.sqrt(np.square(o).sum(axis=0, keepdims=True))
    assert np.array_equal(o, golden["normc_b_out"])


def test_wrap_deepmind_golden(golden, oracle):
    """wrap_deepmind semantics (noop/fire reset, max-and-skip, PIL warp, stack) against the real
    reference wrappers driven over the same SynthAtari frames."""
    tot_px = bad_px = 0
    for s in golden["wrap_seeds"]:
        env = oracle.WrappedEnv()
        ob = env.reset(int(s))
        ref_obs = golden["wrap_s%d_obs" % s]
        diffs = [np.abs(ob.astype(int) - ref_obs[0].astype(int))]
        for t, a in enumerate(golden["wrap_s%d_actions" % s]):
            ob, r, d = env.step(int(a))
            assert r == golden["wrap_s%d_rews" % s][t]
            assert d == bool(golden["wrap_s%d_dones" % s][t])
            assert np.array_equal(env.ram(), golden["wrap_s%d_rams" % s][t])
            diffs.append(np.abs(ob.astype(int) - ref_obs[t + 1].astype(int)))
        dd = np.stack(diffs)
        # gray dot goes through BLAS in the reference (order unpinned): <= 1 LSB on rare pixels
        assert dd.max() <= 1
        tot_px += dd.size; bad_px += int((dd > 0).sum())
    assert bad_px <= 1e-3 * tot_px, (bad_px, tot_px)
    assert np.array_equal(oracle.warp_rgb(
        np.random.RandomState(1).randint(0, 256, (210, 160, 3)).astype(np.uint8)), golden["warp_rs1"])


def test_wrap_early_done(golden, oracle):
    env = oracle.WrappedEnv()
    env.reset(5)
    rews, obs = [], []
    for t in range(400):
        ob, r, d = env.step(5)
        rews.append(r); obs.append(ob)
        if d:
            break
    assert len(rews) == int(golden["wrap_down_len"])
    assert np.array_equal(np.array(rews, np.float32), golden["wrap_down_rews"])
    assert np.abs(obs[-1].astype(int) - golden["wrap_down_last_obs"].astype(int)).max() <= 1


def test_resize_against_pil(oracle):
    from PIL import Image
    rs = np.random.RandomState(11)
    for _ in range(3):
        rgb = rs.randint(0, 256, (210, 160, 3)).astype(np.uint8)
        r = rgb[..., 0].astype(np.float32) * np.float32(0.299)
        g = rgb[..., 1].astype(np.float32) * np.float32(0.587)
        b = rgb[..., 2].astype(np.float32) * np.float32(0.114)
        gray = (r + g) + b
        pil = np.array(Image.fromarray(gray).resize((84, 84), resample=Image.BILINEAR), dtype=np.uint8)
        assert np.array_equal(oracle.warp_rgb(rgb), pil)
    kh, bh, kv, bv = oracle.resize_tables()
    assert np.allclose(kh[0, :3], [0.45864663, 0.42857143, 0.11278196], atol=1e-8)
    assert np.allclose(kv[0, :4], [0.31818181, 0.40909091, 0.22727273, 0.04545455], atol=1e-8)
    assert np.allclose(kh.sum(1), 1) and np.allclose(kv.sum(1), 1)


# ------------------------------------------------------------------ widened rows, pinned to the real reference (reference_wire.npz)
@pytest.fixture(scope="module")
def wire():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_wire.npz"))


def test_reference_pickles_load_as_our_wire_types(wire):
    """Payloads pickled BY the reference (dist.py:19-20) unpickle into dne_hip's wire types on a machine without the reference
    (the GPU box): module path, field order and array contents survive; and our keys are the reference's."""
    import pickle
    from dne_hip import dist, es, es_modified, ga
    task = pickle.loads(wire["wire_task"].tobytes())
    assert type(task) is es.Task and task.timestep_limit == 7 and task.params.dtype == np.float32 and task.ref_batch[0].shape == (84, 84, 4)
    gtask = pickle.loads(wire["wire_gatask"].tobytes())
    assert type(gtask) is ga.GATask and gtask.population == [[1, 2], [3]] and gtask.timestep_limit == 9
    tid, res = pickle.loads(wire["wire_result"].tobytes())
    assert tid == 4 and type(res) is es.Result and res.worker_id == 3 and res.lengths_n2.dtype == np.int32
    tid, mres = pickle.loads(wire["wire_modified_result"].tobytes())
    assert tid == 6 and type(mres) is es_modified.Result and mres.bc_vectors[0][0].shape == (1, 128) and mres.bc_vectors[0][4] == 777
    cfg = pickle.loads(wire["wire_config"].tobytes())
    assert type(cfg) is es.Config and cfg.noise_stdev == 0.02 and cfg.episode_cutoff_mode == 5
    # and the other way round: what we pickle names the reference's modules (so a reference master can load it)
    mine = dist.serialize((4, es.Result(*res)))
    assert b"es_distributed.es" in mine and b"dne_hip" not in mine
    assert [str(k) for k in wire["wire_keys"]] == [dist.EXP_KEY, dist.TASK_ID_KEY, dist.TASK_DATA_KEY, dist.TASK_CHANNEL, dist.RESULTS_KEY, dist.ARCHIVE_KEY]


def test_gpu_tree_schedules_and_genomes_match_the_reference(wire, oracle):
    """helper.py:46-88 schedules and models/base.py:118-149 genomes against values produced by the reference's own classes."""
    from dne_hip import ga_gpu
    lin = ga_gpu.make_schedule({"type": "LinearSchedule", "schedule": 10, "initial_p": 0.01, "final_p": 0.001, "field": "iteration"})
    assert [lin.value(iteration=int(i), timesteps_so_far=0) for i in wire["sched_iterations"]] == wire["sched_linear"].tolist()
    assert ga_gpu.make_schedule(0.002).value(iteration=3) == wire["sched_constant"][0]
    sb = ga_gpu.model_scale_by(18)
    sel = wire["gpu_sel"]
    assert np.array_equal(sb[sel], wire["gpu_scale_by_sel"])
    noise = np.random.RandomState(123).randn(4_000_000).astype(np.float32)
    idx, pw = wire["gpu_seeds_idx"], wire["gpu_seeds_power"]
    genome = (int(idx[0]), (int(idx[1]), float(pw[1])), (int(idx[2]), float(pw[2])))
    for n in range(3):
        th = oracle.ga_gpu_rebuild(noise, genome[:n + 1], sb)
        assert str(wire["gpu_theta%d_dtype" % n]) == "float32"
        assert np.array_equal(th[sel], wire["gpu_theta%d_sel" % n]), n
        assert th.astype(np.float64).sum() == float(wire["gpu_theta%d_sum" % n])
