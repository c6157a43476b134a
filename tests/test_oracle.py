"""CPU tests of the oracle (no GPU): fp64 spec <-> fp32 fast agreement, analytic known-answer
tests of the TF1.12 op semantics (SURVEY.md §8(c) list) and the committed golden vectors."""
import numpy as np
import pytest

from conftest import geometry_of, load_golden
from oracle import pfnl_fast, pfnl_spec
from pfnl_amd import synth
from pfnl_amd.spec import PFNLGeometry


def test_geometry_matches_survey():
    g = PFNLGeometry()
    assert g.num_params() == 3003156
    assert len(g.weight_shapes()) == 130
    assert abs(g.flops_per_clip(128, 128) / 1e9 - 539.50) < 0.01      # §8(d)
    assert abs(g.flops_per_clip(32, 32) / 1e9 - 33.39) < 0.01
    assert abs(g.flops_per_clip(128, 128, shared_base=True) / 1e9 - 394.5) < 0.1


def test_d2s_s2d_roundtrip_and_index_formula():
    rng = np.random.default_rng(0)
    x = rng.random((2, 6, 8, 5))
    assert np.array_equal(pfnl_spec.depth_to_space2(pfnl_spec.space_to_depth2(x)), x)
    z = rng.random((1, 3, 4, 12))
    d = pfnl_spec.depth_to_space2(z)
    for h in range(3):
        for w in range(4):
            for i in range(2):
                for j in range(2):
                    for k in range(3):
                        assert d[0, 2 * h + i, 2 * w + j, k] == z[0, h, w, (2 * i + j) * 3 + k]
    s = pfnl_spec.space_to_depth2(x)
    assert s[1, 2, 3, (1 * 2 + 0) * 5 + 4] == x[1, 2 * 2 + 1, 2 * 3 + 0, 4]


def test_bicubic_legacy_semantics():
    rng = np.random.default_rng(1)
    x = rng.random((1, 5, 7, 3))
    b = pfnl_spec.resize_bicubic_tf1(x, 4)
    assert np.array_equal(b[:, ::4, ::4], x)                # t=0 taps are [0,1,0,0]
    c = pfnl_spec.resize_bicubic_tf1(np.full((1, 4, 4, 3), 0.37), 4)
    assert np.allclose(c, 0.37, atol=1e-15)                 # weights sum to one, clamped border
    w = pfnl_spec._bicubic_taps(0.25)
    assert np.allclose(w, [-0.10546875, 0.87890625, 0.26171875, -0.03515625], atol=0)
    assert np.allclose(pfnl_spec._bicubic_taps(0.5), [-0.09375, 0.59375, 0.59375, -0.09375], atol=0)
    # interior sample = explicit 4x4 tap sum
    oy, ox = 9, 14
    iy, ix, ty, tx = oy // 4, ox // 4, (oy % 4) / 4, (ox % 4) / 4
    wy, wx = pfnl_spec._bicubic_taps(ty), pfnl_spec._bicubic_taps(tx)
    ref = sum(wy[a] * wx[c] * x[0, iy - 1 + a, ix - 1 + c] for a in range(4) for c in range(4))
    assert np.allclose(b[0, oy, ox], ref, atol=1e-14)


def test_conv_delta_kernel_is_shift_and_same_padding():
    rng = np.random.default_rng(2)
    x = rng.random((1, 6, 6, 2))
    k = np.zeros((3, 3, 2, 2))
    k[0, 2, 0, 0] = 1.0      # tap (dy=-1, dx=+1) of channel 0
    k[0, 2, 1, 1] = 1.0
    y = pfnl_spec.conv2d_same(x, k, None)
    assert np.array_equal(y[0, 1:, :-1], x[0, :-1, 1:])     # cross-correlation, no flip
    assert np.all(y[0, 0] == 0) and np.all(y[0, :, -1] == 0)   # zero padding


def _nl_weights(C, rng, zero_g=False):
    wg = np.zeros((1, 1, C, C)) if zero_g else rng.normal(size=(1, 1, C, C)) * 0.1
    return wg, rng.normal(size=C) * 0.1, rng.normal(size=(1, 1, C, C)) * 0.1, rng.normal(size=C) * 0.1


def test_nonlocal_known_answers():
    rng = np.random.default_rng(3)
    C = 12
    # constant input => uniform affinity => Y = mean of G rows = G (all rows equal)
    x = np.full((1, 4, 4, C), 0.3)
    wg, bg, ww, bw = _nl_weights(C, rng)
    z = pfnl_spec.nonlocal_block(x, wg, bg, ww, bw)
    g = x.reshape(-1, C) @ wg[0, 0] + bg
    assert np.allclose(z.reshape(-1, C), g @ ww[0, 0] + bw, atol=1e-13)
    # Wg = 0 => Z = bg Ww + bw broadcast
    x = rng.random((2, 4, 6, C))
    wg, bg, ww, bw = _nl_weights(C, rng, zero_g=True)
    z = pfnl_spec.nonlocal_block(x, wg, bg, ww, bw)
    assert np.allclose(z, bg @ ww[0, 0] + bw, atol=1e-13)
    # fold (P X)(Wg Ww) + (bg Ww + bw) == two-step, and stabilised == as-written softmax
    wg, bg, ww, bw = _nl_weights(C, rng)
    z = pfnl_spec.nonlocal_block(x, wg, bg, ww, bw)
    X = x.reshape(2, -1, C)
    S = X @ X.transpose(0, 2, 1)
    assert np.allclose(S, S.transpose(0, 2, 1))
    P = np.exp(S)
    P /= P.sum(-1, keepdims=True)
    assert np.allclose(P.sum(-1), 1.0)
    fold = (P @ X) @ (wg[0, 0] @ ww[0, 0]) + (bg @ ww[0, 0] + bw)
    assert np.allclose(z.reshape(2, -1, C), fold, atol=1e-12)
    assert np.allclose(pfnl_spec.nonlocal_block(x, wg, bg, ww, bw, stabilise=True), z, atol=1e-12)


def test_zero_weights_give_bicubic_and_shared_base_split():
    g = PFNLGeometry(num_block=2)
    w = synth.synthetic_weights(g)
    x = synth.uniform_clips(1, 7, 8, 12, seed=3)
    zero = {k: np.zeros_like(v) for k, v in w.items()}
    y = pfnl_spec.forward(x, zero, num_block=2)
    assert np.allclose(y[:, 0], pfnl_spec.resize_bicubic_tf1(x[:, 3].astype(np.float64), 4), atol=1e-15)
    # conv2 over concat([base, f]) == conv(base; rows 0..63) + conv(f; rows 64..127)   (pfnl.py:69-70)
    rng = np.random.default_rng(4)
    base, f = rng.random((1, 5, 6, 64)), rng.random((1, 5, 6, 64))
    k = w["nlvsr/conv2_0/kernel"].astype(np.float64)
    b = w["nlvsr/conv2_0/bias"].astype(np.float64)
    full = pfnl_spec.conv2d_same(np.concatenate([base, f], -1), k, b)
    split = pfnl_spec.conv2d_same(base, k[:, :, :64], None) + pfnl_spec.conv2d_same(f, k[:, :, 64:], b)
    assert np.allclose(full, split, atol=1e-13)
    # a PF block with zero conv2 is the identity on inp0
    w2 = dict(w)
    for i in range(2):
        w2[f"nlvsr/conv2_{i}/kernel"] = np.zeros_like(w[f"nlvsr/conv2_{i}/kernel"])
        w2[f"nlvsr/conv2_{i}/bias"] = np.zeros_like(w[f"nlvsr/conv2_{i}/bias"])
    t0, t2 = {}, {}
    pfnl_spec.forward(x, w2, num_block=0, taps=t0)
    pfnl_spec.forward(x, w2, num_block=2, taps=t2)
    assert np.allclose(t0["trunk"], t2["trunk"], atol=1e-15)


def test_fast_oracle_matches_spec_oracle():
    for T, scale, nb, H, W in [(7, 4, 3, 16, 20), (5, 2, 2, 12, 16)]:
        g = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
        w = synth.synthetic_weights(g)
        x = synth.uniform_clips(2, T, H, W, seed=11)
        y64 = pfnl_spec.forward(x, w, scale=scale, num_block=nb)
        y32 = pfnl_fast.FastOracle(w, T, scale, nb).forward(x)
        assert y32.shape == y64.shape == (2, 1, scale * H, scale * W, 3)
        assert np.abs(y32 - y64).max() < 5e-6


@pytest.mark.parametrize("name", ["cfg1_7x32x32", "field_7x32x32", "ragged_7x20x36_nb2", "x2_5x16x24_nb2",
                                  "t3_3x12x40_nb1"])
def test_golden_vectors_reproduce(name):
    """The committed vectors are what the oracles produce today (guards against silent drift)."""
    gd = load_golden(name)
    geom = geometry_of(gd["meta"])
    w = synth.synthetic_weights(geom, seed=0)
    y = pfnl_fast.FastOracle(w, geom.num_frames, geom.scale, geom.num_block).forward(gd["x"])
    assert np.abs(y - gd["y"]).max() < 5e-6
    if name in ("ragged_7x20x36_nb2", "x2_5x16x24_nb2", "t3_3x12x40_nb1"):      # small: also the fp64 spec
        y64 = pfnl_spec.forward(gd["x"], w, scale=geom.scale, num_block=geom.num_block)
        assert np.abs(y64 - gd["y"]).max() < 1e-6


def test_golden_psnr_metric():
    gd = load_golden("field_7x32x32")
    p_sr = synth.psnr(gd["y"][:, 0], gd["gt"])
    p_bic = synth.psnr(gd["bic"], gd["gt"])
    assert 15 < p_sr < 60 and 15 < p_bic < 60


def test_harness_windows_and_quantisation():
    gd = load_golden("harness_5x16x24_nb1")
    lrs = (gd["lr_u8"] / 255.).astype(np.float32)
    win = pfnl_spec.sliding_windows(lrs, 7)
    assert win.shape == (5, 7, 16, 24, 3)
    assert np.array_equal(win[0, :4], np.stack([lrs[0]] * 4))      # clamped at the start
    assert np.array_equal(win[4, 3:], np.stack([lrs[4]] * 4))      # and at the end
    assert np.array_equal(pfnl_spec.quantise(gd["sr"][:, 0]), gd["sr_u8"])
    assert pfnl_spec.quantise(np.array([0.5 / 255, 1.5 / 255, 2.5 / 255, -1.0, 2.0])).tolist() == [0, 2, 2, 0, 255]


def test_bf16_trunk_mode_of_the_fast_oracle():
    """trunk_dtype="bf16" (the build-defined arithmetic of BASELINE.json configs[3]): bf16 stores inside the progressive-fusion
    blocks only perturb the fp32 result at the bf16 level (PSNR > 60 dB, |dPSNR(GT)| << 0.1 dB), every stored trunk value
    is exactly representable in bf16, and with zero trunk weights the mode is the fp32 graph."""
    import torch
    from oracle import pfnl_fast
    from pfnl_amd import synth
    from pfnl_amd.spec import PFNLGeometry
    geom = PFNLGeometry(num_frames=7, scale=4, num_block=4)
    w = synth.synthetic_weights(geom, seed=0)
    x, gt = synth.moving_field_clips(1, 7, 16, 24, 4, seed=9)
    o32 = pfnl_fast.FastOracle(w, 7, 4, 4).forward(x)
    fo16 = pfnl_fast.FastOracle(w, 7, 4, 4, trunk_dtype="bf16")
    o16 = fo16.forward(x)
    assert synth.psnr(o16, o32) > 60.0
    assert abs(synth.psnr(o16[:, 0], gt) - synth.psnr(o32[:, 0], gt)) < 0.05
    assert np.array_equal(fo16.forward(x), o16)
    fr = torch.rand(7, 64, 8, 8)
    out = fo16._trunk_bf16(fr, 1, 7, 64, 8, 8)
    assert torch.equal(out, out.to(torch.bfloat16).to(torch.float32))


def test_bf16_golden_outputs_reproduce():
    """tests/golden/bf16_outputs.npz is what the fast oracle's bf16 mode produces today (guards against silent drift)."""
    import os
    from oracle import pfnl_fast
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_outputs.npz"))
    for name in ("ragged_7x20x36_nb2", "x2_5x16x24_nb2"):
        gd = load_golden(name)
        geom = geometry_of(gd["meta"])
        w = synth.synthetic_weights(geom, seed=0)
        y = pfnl_fast.FastOracle(w, geom.num_frames, geom.scale, geom.num_block, trunk_dtype="bf16").forward(gd["x"])
        assert np.abs(y - ref[name]).max() < 1e-5


def test_nonlocal_options_known_answers():
    """utils.NonLocalBlock's other branches (utils.py:27-28,35-36 sub_sample; :59-62 nltype 2) in the spec: known answers."""
    rng = np.random.default_rng(5)
    C = 12
    mk = lambda sc: rng.normal(size=(1, 1, C, C)) * sc                           # noqa: E731
    wg, ww, wt, wp = mk(0.3), mk(0.3), mk(0.3), mk(0.3)
    bg, bw, bt, bp = (rng.normal(size=C) * 0.1 for _ in range(4))
    # average pooling, 'valid': windows that do not fit are dropped
    a = rng.random((2, 7, 5, 3))
    p = pfnl_spec.avg_pool_valid(a, 2)
    assert p.shape == (2, 3, 2, 3)
    assert np.allclose(p[1, 2, 1], a[1, 4:6, 2:4].mean(axis=(0, 1)))
    # a constant image: every affinity row is uniform whatever the normalisation -> y = g of that pixel, in all three types
    x = np.broadcast_to(rng.random(C), (1, 4, 6, C)).copy()
    outs = [pfnl_spec.nonlocal_block(x, wg, bg, ww, bw, theta=(wt, bt + 2.0), phi=(wp, bp + 2.0), nltype=t, sub_sample=s)
            for t in (0, 1, 2) for s in (1, 2)]
    want = pfnl_spec.conv2d_same(pfnl_spec.conv2d_same(x, wg, bg), ww, bw)
    for o in outs:
        assert np.allclose(o, want, atol=1e-12)
    # an image that is constant on 2x2 blocks: pooling merges equal keys, which changes no softmax / relu-normalised average
    blocks = rng.random((1, 3, 4, C))
    xb = np.repeat(np.repeat(blocks, 2, axis=1), 2, axis=2)
    for t in (0, 1, 2):
        full = pfnl_spec.nonlocal_block(xb, wg, bg, ww, bw, theta=(wt, bt + 1.0), phi=(wp, bp + 1.0), nltype=t, sub_sample=1)
        pooled = pfnl_spec.nonlocal_block(xb, wg, bg, ww, bw, theta=(wt, bt + 1.0), phi=(wp, bp + 1.0), nltype=t, sub_sample=2)
        assert np.allclose(full, pooled, atol=1e-12), t
    # nltype 2 on two keys by hand: P = relu(f) / rowsum
    x2 = rng.random((1, 1, 2, C))
    bt, bp = bt + 1.0, bp + 1.0
    th = pfnl_spec.conv2d_same(x2, wt, bt).reshape(2, C)
    ph = pfnl_spec.conv2d_same(x2, wp, bp).reshape(2, C)
    g = pfnl_spec.conv2d_same(x2, wg, bg).reshape(2, C)
    fm = np.maximum(th @ ph.T, 0)
    y = (fm / fm.sum(1, keepdims=True)) @ g
    want = pfnl_spec.conv2d_same(y.reshape(1, 1, 2, C), ww, bw)
    got = pfnl_spec.nonlocal_block(x2, wg, bg, ww, bw, theta=(wt, bt), phi=(wp, bp), nltype=2)
    assert np.isfinite(want).all() and np.allclose(got, want, atol=1e-12)
    # a query without a positive affinity divides 0 by 0 (utils.py:61-63 has no epsilon): NaN, as TF gives
    neg = pfnl_spec.nonlocal_block(x2, wg, bg, ww, bw, theta=(0 * wt, np.ones(C)), phi=(0 * wp, -np.ones(C)), nltype=2)
    assert np.isnan(neg).all()
    with pytest.raises(ValueError):
        pfnl_spec.nonlocal_block(x2, wg, bg, ww, bw, nltype=3)
