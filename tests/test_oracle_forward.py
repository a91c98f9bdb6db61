"""CPU: cross-check the oracle's restatement of the (absent, unpinned) TensorFlow forward semantics
against an independent torch-CPU fp32 formulation: SAME padding asymmetry, HWIO weights, NHWC flatten
order, batch-norm with biased batch variance, first-max argmax.  Tolerance 1e-4 (summation order)."""
import numpy as np
import pytest

# torch is imported inside the function: pytest imports every test module at collection, and a process that has torch loaded
# resolves libamdhip64.so.7 to torch's bundled ROCm 7.0 runtime -- the GPU tests must run on /opt/rocm's, like bench.py


def _torch_forward(L, th, obs_u8, kind, ref_u8=None):
    import torch
    import torch.nn.functional as F
    th = torch.from_numpy(th)
    A = L.nact

    def t(off, shape):
        n = int(np.prod(shape))
        return th[off:off + n].reshape(shape)

    w1 = t(L.c1w, (8, 8, 4, 16)).permute(3, 2, 0, 1); b1 = t(L.c1b, (16,))
    w2 = t(L.c2w, (4, 4, 16, 32)).permute(3, 2, 0, 1); b2 = t(L.c2b, (32,))
    wf = t(L.fcw, (3872, 256)); bf = t(L.fcb, (256,))
    wo = t(L.ow, (256, A)); bo = t(L.ob, (A,))

    def trunk(x_u8, stats=None, collect=None):
        x = torch.from_numpy(x_u8.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2)
        y1 = F.conv2d(F.pad(x, (2, 2, 2, 2)), w1, b1, stride=4)          # SAME: 2/2
        a1 = bn(y1, 1, stats, collect)
        y2 = F.conv2d(F.pad(a1, (1, 2, 1, 2)), w2, b2, stride=2)         # SAME: 1 before, 2 after
        a2 = bn(y2, 2, stats, collect)
        flat = a2.permute(0, 2, 3, 1).reshape(x.shape[0], 3872)          # NHWC flatten
        y3 = flat @ wf + bf
        a3 = bn(y3, 3, stats, collect)
        return y1, y2, y3, a3 @ wo + bo

    def bn(y, i, stats, collect):
        if kind == 1:
            return torch.relu(y)
        beta = t(getattr(L, "bn%db" % i), (y.shape[1],)); gamma = t(getattr(L, "bn%dg" % i), (y.shape[1],))
        dims = [0, 2, 3] if y.dim() == 4 else [0]
        if collect is not None:
            mean = y.mean(dims); var = y.var(dims, unbiased=False)
            collect[i] = (mean, var)
        else:
            mean, var = stats[i]
        shp = [1, -1, 1, 1] if y.dim() == 4 else [1, -1]
        out = (y - mean.reshape(shp)) * (gamma / torch.sqrt(var + 1e-3)).reshape(shp) + beta.reshape(shp)
        return torch.relu(out)

    stats = None
    if kind == 0:
        stats = {}
        trunk(ref_u8, collect=stats)
    return [v.numpy() for v in trunk(obs_u8, stats=stats)], stats


@pytest.mark.parametrize("kind,nact", [(0, 18), (1, 18), (0, 14)])
def test_forward_vs_torch(oracle, kind, nact):
    O = oracle
    L = O.layout(kind, nact)
    rs = np.random.RandomState(7 + kind)
    if kind == 0:
        th = O.es_init_theta(L, 0)
        th += (0.02 * rs.randn(L.P)).astype(np.float32)  # perturbed betas/gammas/biases too
    else:
        th = O.ga_normc(L, rs.randn(L.P).astype(np.float32))
        th += (0.005 * rs.randn(L.P)).astype(np.float32)
    ref = O.get_ref_batch(seed=0, batch_size=16, nact=nact)
    obs = rs.randint(0, 256, (4, 84, 84, 4)).astype(np.uint8)
    obs[1] = ref[3]
    bn = O.es_ref_pass(L, th, ref) if kind == 0 else None
    (y1, y2, y3, lg), stats = _torch_forward(L, th, obs, kind, ref)
    for i in range(obs.shape[0]):
        o1, o2, o3, ol = O.forward_debug(L, th, bn, obs[i])
        assert np.allclose(o1.reshape(21, 21, 16), y1[i].transpose(1, 2, 0), atol=1e-4, rtol=1e-4)
        assert np.allclose(o2.reshape(11, 11, 32), y2[i].transpose(1, 2, 0), atol=2e-4, rtol=1e-4)
        assert np.allclose(o3, y3[i], atol=5e-4, rtol=1e-4)
        assert np.allclose(ol, lg[i], atol=5e-4, rtol=1e-4)
        a, _ = O.act(L, th, bn, obs[i])
        srt = np.sort(lg[i])
        if srt[-1] - srt[-2] > 1e-3:
            assert a == int(np.argmax(lg[i]))
    if kind == 0:
        # scale/shift agree with torch's batch moments
        for i, (o, C) in enumerate(((0, 16), (32, 32), (96, 256)), 1):
            mean, var = stats[i]
            gamma = th[getattr(L, "bn%dg" % i):][:C]; beta = th[getattr(L, "bn%db" % i):][:C]
            sc = gamma / np.sqrt(var.numpy() + 1e-3)
            assert np.allclose(bn[o:o + C], sc, rtol=1e-4, atol=1e-6)
            assert np.allclose(bn[o + C:o + 2 * C], beta - mean.numpy() * sc, rtol=1e-3, atol=1e-4)


def test_argmax_first_max(oracle):
    O = oracle
    L = O.layout(O.KIND_GA, 18)
    th = np.zeros(L.P, np.float32)  # all logits equal -> first index (tf.argmax)
    a, lg = O.act(L, th, None, np.zeros((84, 84, 4), np.uint8))
    assert a == 0 and not lg.any()
    th[L.ob + 5] = 1.0; th[L.ob + 9] = 1.0
    a, _ = O.act(L, th, None, np.zeros((84, 84, 4), np.uint8))
    assert a == 5


def test_antithetic_symmetry(oracle, small_noise):
    # gpu_implementation/es.py:182-183: the reference's only numeric tolerance
    O = oracle
    L = O.layout(O.KIND_ES)
    th = O.es_init_theta(L, 0)
    idx = 12345
    p = O.perturb(th, small_noise, idx, 0.02, +1); n = O.perturb(th, small_noise, idx, 0.02, -1)
    assert np.abs((p + n) / 2 - th).max() < 1e-5
    v = np.float32(0.02) * small_noise[idx:idx + L.P]
    assert np.array_equal(p, th + v) and np.array_equal(n, th - v)


def test_large_model_forward_vs_torch(oracle):
    """LargeModel of the GPU tree (gpu_implementation/neuroevolution/models/dqn.py:39-47 over models/base.py:50-95: SAME
    patches x reshaped weights + bias, relu): conv 32 8x8/4, conv 64 4x4/2, conv 64 3x3/1, fc 512, out.  TensorFlow is absent
    ("parity unpinned"), so the oracle's restatement is cross-checked against torch like the small networks above."""
    import torch
    import torch.nn.functional as F
    O = oracle
    L = O.layout(O.KIND_GA_LARGE, 18)
    assert L.P == 8 * 8 * 4 * 32 + 32 + 4 * 4 * 32 * 64 + 64 + 3 * 3 * 64 * 64 + 64 + 7744 * 512 + 512 + 512 * 18 + 18 == 4052658
    rs = np.random.RandomState(3)
    th = (rs.randn(L.P) * 0.03).astype(np.float32)
    t = torch.from_numpy(th)

    def tt(off, shape):
        return t[off:off + int(np.prod(shape))].reshape(shape)
    w1 = tt(L.c1w, (8, 8, 4, 32)).permute(3, 2, 0, 1); b1 = tt(L.c1b, (32,))
    w2 = tt(L.c2w, (4, 4, 32, 64)).permute(3, 2, 0, 1); b2 = tt(L.c2b, (64,))
    w3 = tt(L.c3w, (3, 3, 64, 64)).permute(3, 2, 0, 1); b3 = tt(L.c3b, (64,))
    wf = tt(L.fcw, (7744, 512)); bf = tt(L.fcb, (512,)); wo = tt(L.ow, (512, 18)); bo = tt(L.ob, (18,))
    obs = rs.randint(0, 256, (3, 84, 84, 4)).astype(np.uint8)
    x = torch.from_numpy(obs.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2)
    y1 = F.conv2d(F.pad(x, (2, 2, 2, 2)), w1, b1, stride=4)                      # SAME: total 4 -> 2 / 2
    y2 = F.conv2d(F.pad(torch.relu(y1), (1, 2, 1, 2)), w2, b2, stride=2)         # total 3 -> 1 before, 2 after
    y3 = F.conv2d(F.pad(torch.relu(y2), (1, 1, 1, 1)), w3, b3, stride=1)         # total 2 -> 1 / 1
    y4 = torch.relu(y3).permute(0, 2, 3, 1).reshape(3, 7744) @ wf + bf            # NHWC flatten (models/base.py:97-98)
    lg = torch.relu(y4) @ wo + bo
    for i in range(3):
        o1, o2, o3, o4, ol = O.forward_large_debug(L, th, obs[i])
        assert np.allclose(o1.reshape(21, 21, 32), y1[i].permute(1, 2, 0).numpy(), atol=1e-4, rtol=1e-4)
        assert np.allclose(o2.reshape(11, 11, 64), y2[i].permute(1, 2, 0).numpy(), atol=2e-4, rtol=1e-4)
        assert np.allclose(o3.reshape(11, 11, 64), y3[i].permute(1, 2, 0).numpy(), atol=5e-4, rtol=1e-4)
        assert np.allclose(o4, y4[i].numpy(), atol=2e-3, rtol=1e-4)
        assert np.allclose(ol, lg[i].numpy(), atol=2e-3, rtol=1e-4)
        a, lg2 = O.act(L, th, None, obs[i])
        assert a == int(np.argmax(ol)) and np.array_equal(lg2, ol)


def _conv64(x, w, b, stride, pad_lo, pad_hi):
    """float64 SAME-style convolution, NHWC x HWIO, of ONE image: returns (sum, sum of |terms|) per output element"""
    H, W, C = x.shape
    kh, kw, _, CO = w.shape
    xp = np.zeros((H + pad_lo + pad_hi, W + pad_lo + pad_hi, C), np.float64)
    xp[pad_lo:pad_lo + H, pad_lo:pad_lo + W] = x
    oh = (xp.shape[0] - kh) // stride + 1
    ow = (xp.shape[1] - kw) // stride + 1
    cols = np.empty((oh, ow, kh, kw, C), np.float64)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, i, j, :] = xp[i:i + stride * oh:stride, j:j + stride * ow:stride, :]
    cols = cols.reshape(oh * ow, kh * kw * C)
    wm = w.reshape(kh * kw * C, CO).astype(np.float64)
    return cols @ wm + b, np.abs(cols) @ np.abs(wm) + np.abs(b)


@pytest.mark.parametrize("kind,nact", [(0, 18), (1, 18)])
def test_every_layer_is_a_valid_fp32_evaluation_of_the_float64_layer(oracle, kind, nact):
    """TensorFlow 0.12's fp32 summation order is unknowable here (TF is absent: SURVEY 8c), so no test can pin the forward pass bit for bit.
    What CAN be pinned: every layer of the oracle, fed the oracle's own fp32 input of that layer, lies within the worst-case rounding
    bound of ANY fp32 summation order around the float64 value of the TF-defined layer (SAME padding 2/2 and 1/2, HWIO weights, NHWC
    flatten, bias, `y * scale + shift` batch norm, relu):  |fl(sum) - sum| <= gamma_n * sum |x_k w_k|,  gamma_n = n u / (1 - n u), u = 2^-24,
    n = terms + bias (Higham, Accuracy and Stability of Numerical Algorithms, 2nd ed., section 4.2).  An indexing, padding or layout error moves an
    output by a whole term, orders of magnitude beyond the bound."""
    O = oracle
    L = O.layout(kind, nact)
    rs = np.random.RandomState(11 + kind)
    if kind == 0:
        th = O.es_init_theta(L, 0)
        th += (0.02 * rs.randn(L.P)).astype(np.float32)
    else:
        th = O.ga_normc(L, rs.randn(L.P).astype(np.float32))
        th += (0.005 * rs.randn(L.P)).astype(np.float32)
    ref = O.get_ref_batch(seed=0, batch_size=16, nact=nact)
    bn = O.es_ref_pass(L, th, ref) if kind == 0 else None
    u = 2.0 ** -24

    def gamma(n):
        return n * u / (1.0 - n * u)

    def t(off, shape):
        return th[off:off + int(np.prod(shape))].reshape(shape)

    def act(y, o, C):   # the consumer's own fp32 operations: relu(fl(fl(y * scale) + shift)) / relu(y)
        if bn is None:
            return np.maximum(y, np.float32(0))
        s = bn[o:o + C].astype(np.float32); h = bn[o + C:o + 2 * C].astype(np.float32)
        v = (y.astype(np.float32) * s).astype(np.float32)
        v = (v + h).astype(np.float32)
        return np.maximum(v, np.float32(0))

    w1 = t(L.c1w, (8, 8, 4, 16)); b1 = t(L.c1b, (16,)).astype(np.float64)
    w2 = t(L.c2w, (4, 4, 16, 32)); b2 = t(L.c2b, (32,)).astype(np.float64)
    wf = t(L.fcw, (3872, 256)).astype(np.float64); bf = t(L.fcb, (256,)).astype(np.float64)
    wo = t(L.ow, (256, nact)).astype(np.float64); bo = t(L.ob, (nact,)).astype(np.float64)
    obs = rs.randint(0, 256, (3, 84, 84, 4)).astype(np.uint8)
    obs[1] = ref[5]
    for i in range(obs.shape[0]):
        o1, o2, o3, ol = O.forward_debug(L, th, bn, obs[i])
        x = (obs[i].astype(np.float32) / np.float32(255.0)).astype(np.float32)
        y1, m1 = _conv64(x.astype(np.float64), w1, b1, 4, 2, 2)
        assert y1.shape == (441, 16)
        assert (np.abs(o1.reshape(441, 16) - y1) <= gamma(256 + 1) * m1 + 1e-30).all()
        a1 = act(o1.reshape(441, 16), 0, 16).reshape(21, 21, 16)
        y2, m2 = _conv64(a1.astype(np.float64), w2, b2, 2, 1, 2)
        assert y2.shape == (121, 32)
        assert (np.abs(o2.reshape(121, 32) - y2) <= gamma(256 + 1) * m2 + 1e-30).all()
        a2 = act(o2.reshape(121, 32), 32, 32).reshape(3872).astype(np.float64)   # NHWC flatten: position-major, channel-minor
        y3 = a2 @ wf + bf; m3 = np.abs(a2) @ np.abs(wf) + np.abs(bf)
        assert (np.abs(o3 - y3) <= gamma(3872 + 4) * m3 + 1e-30).all()            # + the three adds that combine the four k-slices
        a3 = act(o3, 96, 256).astype(np.float64)
        yl = a3 @ wo + bo; ml = np.abs(a3) @ np.abs(wo) + np.abs(bo)
        assert (np.abs(ol - yl) <= gamma(256 + 4) * ml + 1e-30).all()
        # and the bound is tight enough to mean something: a single dropped or misplaced term would break it
        assert gamma(3872 + 4) * m3.max() < 0.05 * np.abs(a2).max() * np.abs(wf).max() + 1e-3


def test_large_model_layers_within_the_fp32_bound_of_float64(oracle):
    """the same pin for the GPU tree's LargeModel (dqn.py:39-47): every layer inside the any-order fp32 rounding band around float64"""
    O = oracle
    L = O.layout(O.KIND_GA_LARGE, 18)
    rs = np.random.RandomState(5)
    th = (rs.randn(L.P) * 0.03).astype(np.float32)
    u = 2.0 ** -24

    def gamma(n):
        return n * u / (1.0 - n * u)

    def t(off, shape):
        return th[off:off + int(np.prod(shape))].reshape(shape)
    w1 = t(L.c1w, (8, 8, 4, 32)); b1 = t(L.c1b, (32,)).astype(np.float64)
    w2 = t(L.c2w, (4, 4, 32, 64)); b2 = t(L.c2b, (64,)).astype(np.float64)
    w3 = t(L.c3w, (3, 3, 64, 64)); b3 = t(L.c3b, (64,)).astype(np.float64)
    wf = t(L.fcw, (7744, 512)).astype(np.float64); bf = t(L.fcb, (512,)).astype(np.float64)
    wo = t(L.ow, (512, 18)).astype(np.float64); bo = t(L.ob, (18,)).astype(np.float64)
    relu = lambda y: np.maximum(y.astype(np.float32), np.float32(0)).astype(np.float64)
    for obs in rs.randint(0, 256, (2, 84, 84, 4)).astype(np.uint8):
        o1, o2, o3, o4, ol = O.forward_large_debug(L, th, obs)
        x = (obs.astype(np.float32) / np.float32(255.0)).astype(np.float64)
        y, m = _conv64(x, w1, b1, 4, 2, 2)
        assert (np.abs(o1.reshape(441, 32) - y) <= gamma(256 + 1) * m + 1e-30).all()
        y, m = _conv64(relu(o1).reshape(21, 21, 32), w2, b2, 2, 1, 2)
        assert (np.abs(o2.reshape(121, 64) - y) <= gamma(512 + 1) * m + 1e-30).all()
        y, m = _conv64(relu(o2).reshape(11, 11, 64), w3, b3, 1, 1, 1)
        assert (np.abs(o3.reshape(121, 64) - y) <= gamma(576 + 1) * m + 1e-30).all()
        a = relu(o3).reshape(7744)
        y = a @ wf + bf; m = np.abs(a) @ np.abs(wf) + np.abs(bf)
        assert (np.abs(o4 - y) <= gamma(7744 + 8) * m + 1e-30).all()
        a = relu(o4)
        y = a @ wo + bo; m = np.abs(a) @ np.abs(wo) + np.abs(bo)
        assert (np.abs(ol - y) <= gamma(512 + 8) * m + 1e-30).all()
