"""bf16 trunk (option precision=bf16, BASELINE.json configs[3]) on a real MI355X: the bf16 kernels against a torch-CPU
restatement with the same rounding points (bf16 operands, fp32 accumulation, bf16 stores), through the C-ABI.

Tolerances.  A bf16 store rounds to nearest even: |error| <= 2^-9 |v| per element; the fp32 accumulation order of
the MFMA differs from the CPU's, which can move a value across a rounding boundary, i.e. up to one bf16 ulp
(2^-8 |v|) on isolated elements.  Ops: |d| <= 2^-7 |ref| + 1e-5 element-wise and a mean error far below one ulp.
Whole forward (20 blocks of such stores, then the fp32 merge / tail): PSNR against the equally-rounded oracle
> 55 dB and |dPSNR| against ground truth vs the fp32 oracle reported and bounded at 0.1 dB (NOT the fp32 bar)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from oracle import pfnl_fast  # noqa: E402
from pfnl_amd import ops, synth  # noqa: E402
from pfnl_amd.engine import PFNLEngine  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def close_bf16(out, ref32):
    d = (out.float().cpu() - ref32).abs()
    bound = ref32.abs() * 2.0 ** -7 + 1e-5
    assert bool((d <= bound).all()), float((d - bound).max())
    assert float(d.mean()) <= float(ref32.abs().mean()) * 2.0 ** -9 + 1e-6


def ref_conv3(x, k, b, act, addend=None, add_div=1, resid=None):
    y = F.conv2d(x.permute(0, 3, 1, 2), r16(torch.from_numpy(k)).permute(3, 2, 0, 1), torch.from_numpy(b), padding=1)
    y = y.permute(0, 2, 3, 1)
    if addend is not None:
        y = y + addend.repeat_interleave(add_div, 0)
    if act:
        y = F.leaky_relu(y, 0.2)
    if resid is not None:
        y = y + resid
    return y


@pytest.mark.parametrize("shape", [(3, 20, 36), (2, 16, 32), (1, 50, 70), (2, 2, 2), (1, 33, 95), (7, 18, 34)])
@pytest.mark.parametrize("act", [True, False])
def test_conv3x3_bf16_plain(shape, act):
    Fr, H, W = shape
    g = torch.Generator().manual_seed(Fr * 1000 + H)
    x = r16(torch.randn(Fr, H, W, 64, generator=g))
    k = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
    b = (torch.randn(64, generator=g) * 0.1).numpy()
    out = ops.conv3x3_bf16(x.to(torch.bfloat16).cuda(), k, b, act=act)
    close_bf16(out, ref_conv3(x, k, b, act))


@pytest.mark.parametrize("clips,T,H,W", [(2, 7, 20, 36), (1, 5, 32, 64), (3, 3, 17, 33)])
def test_conv3x3_bf16_fused(clips, T, H, W):
    g = torch.Generator().manual_seed(clips * 100 + T)
    Fr = clips * T
    x = r16(torch.randn(Fr, H, W, 64, generator=g))
    addend = r16(torch.randn(clips, H, W, 64, generator=g))
    resid = r16(torch.randn(Fr, H, W, 64, generator=g))
    k = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
    b = (torch.randn(64, generator=g) * 0.1).numpy()
    out = ops.conv3x3_bf16(x.to(torch.bfloat16).cuda(), k, b, act=True, addend=addend.to(torch.bfloat16).cuda(), add_div=T,
                           resid=resid.to(torch.bfloat16).cuda())
    close_bf16(out, ref_conv3(x, k, b, True, addend, T, resid))


@pytest.mark.parametrize("items,T,H,W", [(2, 7, 20, 36), (3, 5, 9, 31), (1, 3, 64, 64), (2, 7, 1, 5)])
def test_conv1x1_bf16(items, T, H, W):
    g = torch.Generator().manual_seed(items * 10 + T)
    x = r16(torch.randn(items * T, H, W, 64, generator=g))
    k = (torch.randn(1, 1, 64 * T, 64, generator=g) * 0.05).numpy()
    b = (torch.randn(64, generator=g) * 0.1).numpy()
    out = ops.conv1x1_bf16(x.to(torch.bfloat16).cuda(), k, b, act=True, frames_per_item=T)
    cat = x.reshape(items, T, H, W, 64).permute(0, 2, 3, 1, 4).reshape(items, H, W, T * 64)
    ref = F.leaky_relu(cat @ r16(torch.from_numpy(k[0, 0])) + torch.from_numpy(b), 0.2)
    close_bf16(out, ref)


@pytest.mark.parametrize("geom,B,H,W", [(PFNLGeometry(num_frames=7, scale=4, num_block=20), 1, 32, 32),
                                        (PFNLGeometry(num_frames=7, scale=4, num_block=3), 2, 20, 36),
                                        (PFNLGeometry(num_frames=5, scale=2, num_block=2), 1, 16, 24),
                                        (PFNLGeometry(num_frames=3, scale=4, num_block=2), 2, 34, 70)])
def test_forward_bf16_matches_rounded_oracle(geom, B, H, W):
    w = synth.synthetic_weights(geom, seed=0)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(w)
    eng.set_option("precision", "bf16")
    x, gt = synth.moving_field_clips(B, geom.num_frames, H, W, geom.scale, seed=5)
    y = eng.forward(x)
    o16 = pfnl_fast.FastOracle(w, geom.num_frames, geom.scale, geom.num_block, trunk_dtype="bf16").forward(x)
    o32 = pfnl_fast.FastOracle(w, geom.num_frames, geom.scale, geom.num_block).forward(x)
    assert y.shape == o16.shape and y.dtype == np.float32
    p_same = synth.psnr(y, o16)
    p_fp32 = synth.psnr(y, o32)
    d_gt = abs(synth.psnr(y[:, 0], gt) - synth.psnr(o32[:, 0], gt))
    print(f"bf16 forward {B}x{geom.num_frames}x{H}x{W} nb{geom.num_block}: PSNR vs bf16 oracle {p_same:.1f} dB, "
          f"vs fp32 oracle {p_fp32:.1f} dB, |dPSNR(GT)| {d_gt:.4f} dB, max|d| {np.abs(y - o16).max():.2e}")
    assert p_same > 55.0, p_same
    assert d_gt <= 0.1, d_gt
    eng.set_option("precision", "fp32")                                   # and back: the fp32 path is untouched
    assert np.abs(eng.forward(x) - o32).max() < 2e-4
    eng.close()


@pytest.mark.parametrize("B,H,W,nb", [(1, 32, 32, 20), (1, 100, 130, 2), (2, 96, 96, 2), (1, 144, 180, 1)])
def test_forward_bf16_mid_shapes_take_the_per_tile_structure(B, H, W, nb):
    """Below 136 (clip, tile) chains the bf16 trunk runs conv10_i as its own launch and the per-frame half of conv2_i in flat tile order
    (option split16_mid=auto, capi.hip "MID shapes"; the chained launches leave most CUs idle there: 1x7x32x32 1.37 -> 0.79 ms, 1x7x128x128
    1.46 -> 1.06).  The flat order changes no bit (bf16_conv10=separate + split16_mid=off is the same arithmetic in chain order); the
    fused structure differs from both only by conv10_i's summation order."""
    geom = PFNLGeometry(num_block=nb)
    w = synth.synthetic_weights(geom, seed=0)
    x, _ = synth.moving_field_clips(B, 7, H, W, 4, seed=7)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(w)
    eng.set_option("precision", "bf16")
    y = eng.forward(x)                                             # split16_mid=auto (default)
    assert np.array_equal(y, eng.forward(x))
    eng.set_option("split16_mid", "off")
    y_fused = eng.forward(x)                                       # conv1_i + conv10_i in one launch, chains
    eng.set_option("bf16_conv10", "separate")
    y_sep = eng.forward(x)                                         # the mid structure's launches, chain order
    eng.close()
    assert np.array_equal(y, y_sep)
    o16 = pfnl_fast.FastOracle(w, 7, 4, nb, trunk_dtype="bf16").forward(x)
    assert synth.psnr(y, o16) > 55.0 and synth.psnr(y_fused, o16) > 55.0
    assert synth.psnr(y, y_fused) > 55.0


@pytest.mark.parametrize("T,scale,nb,H,W,Bs", [(7, 4, 2, 128, 128, (4, 5, 6, 9)), (7, 4, 2, 100, 130, (5,)), (5, 2, 2, 96, 128, (7,)), (3, 4, 2, 90, 98, (11,))])
def test_forward_bf16_split_chains(T, scale, nb, H, W, Bs):
    """Round 6: split chains in the bf16 trunk (option split16_splitchains, conv_bf16_v3.hip SPLIT; reference model/pfnl.py:44, 55, 65-71).  A batch whose
    (clip, tile) chains are not a whole number of rounds of the grid ran its last round at the price of a whole one (5 clips of 128x128: 1.52x the
    time of 4); the chains of that round are cut by frames - conv1_i + conv10_i leaves a part's raw conv10_i sum, c10_finalize_bf16_kernel adds the parts
    (+ bias, leaky-relu, the bf16 rounding point of the uncut launch), the per-frame half fetches the addend per part.  Every batch: repeatable bit
    for bit, against the oracle with the same rounding points, and within bf16 rounding of the uncut launch (conv10_i's summation order moves the fp32
    sum in front of its rounding); whole rounds keep plan and bits."""
    geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
    w = synth.synthetic_weights(geom, seed=T)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(w)
    eng.set_option("precision", "bf16")
    fo = pfnl_fast.FastOracle(w, T, scale, nb, trunk_dtype="bf16")
    chains_clip = ((W + 31) // 32) * ((H + 7) // 8)
    for B in Bs:
        x, _ = synth.moving_field_clips(B, T, H, W, scale, seed=3)
        pl = eng.plan(B, H, W)
        R = (B * chains_clip) % 256
        cut = B * chains_clip > 256 and 0 < R <= 128 and min(T, 256 // R) >= 2
        assert pl["structure"] == ("bf16_3_split" if cut else "bf16_3"), (B, pl)
        y = eng.forward(x)
        assert np.array_equal(y, eng.forward(x)), B
        eng.set_option("split16_splitchains", "off")
        assert eng.plan(B, H, W)["structure"] == "bf16_3"
        y_off = eng.forward(x)
        eng.set_option("split16_splitchains", "auto")
        if cut:
            assert pl["launches_per_block"] == 4 and pl["c1x1"] == 1 and pl["split_parts"] * pl["part_frames"] >= T
            assert synth.psnr(y, y_off) > 60.0, (B, synth.psnr(y, y_off))
        else:
            assert np.array_equal(y.view(np.uint32), y_off.view(np.uint32)), B
        assert synth.psnr(y, fo.forward(x)) > 55.0, B
    eng.close()


@pytest.mark.parametrize("T,scale,nb,B,H,W", [(7, 4, 3, 4, 128, 128), (7, 4, 2, 5, 128, 128), (5, 2, 2, 2, 130, 98), (3, 4, 2, 1, 270, 480), (7, 4, 2, 1, 64, 96)])
def test_forward_bf16_mfma_forms(T, scale, nb, B, H, W):
    """Round 6 (late): the two chained 3x3 launches of the bf16 trunk (reference model/pfnl.py:49-51, 65-71) on v_mfma_f32_16x16x32_bf16 (option bf16_mfma=16, the
    default: the K = 32 shape costs 14 % less energy per FLOP under the package power cap) against the 32x32x16 form (=32): another summation
    order inside the same rounding points - each repeatable bit for bit, each within the tolerance of the bf16 oracle, and within bf16 rounding of each other;
    shapes with whole rounds of chains, split chains, ragged tiles, T = 3 / 5, 2x, and a mid shape (per-tile launches: the option changes nothing there)."""
    geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
    w = synth.synthetic_weights(geom, seed=T)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(w)
    eng.set_option("precision", "bf16")
    assert eng.get_option("bf16_mfma") == "16"
    x, _ = synth.moving_field_clips(B, T, H, W, scale, seed=5)
    y16 = eng.forward(x)
    assert np.array_equal(y16, eng.forward(x))
    eng.set_option("bf16_mfma", "32")
    y32 = eng.forward(x)
    assert np.array_equal(y32, eng.forward(x))
    eng.set_option("bf16_mfma", "16")
    ref = pfnl_fast.FastOracle(w, T, scale, nb, trunk_dtype="bf16").forward(x)
    assert synth.psnr(y16, ref) > 55.0 and synth.psnr(y32, ref) > 55.0, (synth.psnr(y16, ref), synth.psnr(y32, ref))
    assert synth.psnr(y16, y32) > 60.0, synth.psnr(y16, y32)
    with pytest.raises(Exception):
        eng.set_option("bf16_mfma", "8")
    eng.close()


def test_forward_bf16_1080p_against_oracle_subsample():
    """BASELINE.json configs[3] (1080p, bf16) against the ORACLE with the same rounding points
    (oracle/pfnl_fast.py trunk_dtype="bf16"): every 8th HR pixel + a dense 64x64 crop, generated once in the build
    container (tools/make_golden_1080p.py); and against the fp32 oracle subsample."""
    from conftest import load_golden
    gd = load_golden("cfg4_1080p_stride8")
    seed, stride, cy, cx, cs = (int(v) for v in gd["meta"])
    geom = PFNLGeometry()
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(synth.synthetic_weights(geom, seed=0))
    eng.set_option("precision", "bf16")
    y = eng.forward(synth.uniform_clips(1, 7, 270, 480, seed=seed))[0, 0]
    sub, crop = y[::stride, ::stride], y[cy:cy + cs, cx:cx + cs]
    p_same = min(synth.psnr(sub, gd["y_bf16mode"]), synth.psnr(crop, gd["y_bf16mode_crop"]))
    p_fp32 = min(synth.psnr(sub, gd["y_fp32"]), synth.psnr(crop, gd["y_fp32_crop"]))
    print(f"1080p bf16: PSNR vs bf16-mode oracle {p_same:.1f} dB, vs fp32 oracle {p_fp32:.1f} dB, "
          f"max|d| {np.abs(sub - gd['y_bf16mode']).max():.2e}")
    assert p_same > 60.0, p_same
    assert p_fp32 > 55.0, p_fp32
    eng.close()


@pytest.mark.parametrize("B,H,W", [(4, 128, 128), (1, 270, 480)])
def test_forward_bf16_full_size_against_fp32_build(B, H, W):
    """BASELINE.json configs[1] / configs[3] sizes: the CPU oracle needs minutes there, so the bf16 trunk is held against
    the fp32 build of the same forward (itself pinned to the oracle: test_gpu_forward.py, tools/check_1080p.py), plus
    determinism and batch-permutation equivariance (clips are independent units)."""
    geom = PFNLGeometry()
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(synth.synthetic_weights(geom, seed=0))
    x, gt = synth.moving_field_clips(B, 7, H, W, 4, seed=11) if H <= 128 else (synth.uniform_clips(B, 7, H, W, seed=11), None)
    y32 = eng.forward(x)
    eng.set_option("precision", "bf16")
    y16 = eng.forward(x)
    assert y16.shape == y32.shape and np.isfinite(y16).all()
    p = synth.psnr(y16, y32)
    print(f"bf16 vs fp32 build at {B}x7x{H}x{W}: PSNR {p:.1f} dB, max|d| {np.abs(y16 - y32).max():.2e}")
    assert p > 55.0, p
    if gt is not None:
        d = abs(synth.psnr(y16[:, 0], gt) - synth.psnr(y32[:, 0], gt))
        assert d <= 0.1, d
    assert np.array_equal(eng.forward(x), y16)
    if B > 1:
        perm = np.arange(B)[::-1].copy()
        assert np.array_equal(eng.forward(np.ascontiguousarray(x[perm])), y16[perm])
    eng.close()


@pytest.mark.parametrize("precision", ["f16"])
@pytest.mark.parametrize("B,T,H,W", [(1, 7, 16, 16), (2, 7, 20, 36), (1, 5, 18, 22), (1, 3, 34, 30), (1, 7, 64, 64)])
def test_nonlocal_bf16_precision_kernel(B, T, H, W, precision):
    """The non-local kernel of precision=bf16 against the fp64 spec: nonlocal_f16.hip on the hi parts only - binary16 operands
    throughout, 11 mantissa bits in the logits' inputs AND in the probabilities, fp32 accumulation and softmax state.  Bound:
    1e-3 on [0,1]-scale outputs (fp32 kernel: 2e-5); observed 1e-4 ... 5e-4.  (Round 1's split-operand bf16 kernel left the
    library in round 4: tools/experiments/nonlocal_bf16.hip.)"""
    from oracle import pfnl_spec
    rng = np.random.default_rng(B + T + H + W)
    C = 12 * T
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    wg = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    ww = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    bg = rng.normal(size=C).astype(np.float32) * 0.1
    bw = rng.normal(size=C).astype(np.float32) * 0.1
    got = ops.nonlocal_residual(torch.from_numpy(x).cuda(), wg, bg, ww, bw, precision=precision).cpu().numpy()
    x64 = x.astype(np.float64)
    stack = np.concatenate([x64[:, t] for t in range(T)], -1)
    z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64),
                                 ww.astype(np.float64), bw.astype(np.float64), stabilise=True)
    ref = stack + pfnl_spec.depth_to_space2(z)
    err = np.abs(got - ref).max()
    print(f"nonlocal {precision} {B}x{T}x{H}x{W}: max err {err:.2e}")
    assert got.shape == ref.shape and err < 1e-3, err


@pytest.mark.parametrize("precision", ["f16"])
def test_nonlocal_bf16_constant_and_peaked_inputs(precision):
    """Known answers (as for the fp32 kernel): constant frames -> uniform affinity -> Z = (mean G) Ww + bw exactly
    representable path; a bright block -> dominant late keys exercise the running-max rescale with logits ~84."""
    from oracle import pfnl_spec
    T, H, W = 7, 16, 16
    C = 12 * T
    rng = np.random.default_rng(0)
    wg = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    ww = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    bg = rng.normal(size=C).astype(np.float32) * 0.1
    bw = rng.normal(size=C).astype(np.float32) * 0.1
    x = np.full((1, T, H, W, 3), 0.25, np.float32)
    got = ops.nonlocal_residual(torch.from_numpy(x).cuda(), wg, bg, ww, bw, precision=precision).cpu().numpy()
    g = np.full(C, 0.25) @ wg[0, 0].astype(np.float64) + bg
    zc = g @ ww[0, 0].astype(np.float64) + bw
    ref = 0.25 + pfnl_spec.depth_to_space2(np.broadcast_to(zc, (1, H // 2, W // 2, C)).copy())
    assert np.abs(got - ref).max() < 1e-5
    x = (rng.random((1, T, H, W, 3)) * 0.15).astype(np.float32)
    x[:, :, 10:14, 4:8] = 0.97 + 0.03 * rng.random((1, T, 4, 4, 3)).astype(np.float32)
    got = ops.nonlocal_residual(torch.from_numpy(x).cuda(), wg, bg, ww, bw, precision=precision).cpu().numpy()
    x64 = x.astype(np.float64)
    stack = np.concatenate([x64[:, t] for t in range(T)], -1)
    z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64),
                                 ww.astype(np.float64), bw.astype(np.float64), stabilise=True)
    ref = stack + pfnl_spec.depth_to_space2(z)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 1e-3


def test_bf16_kernels_random_geometries_short():
    """tools/stress_bf16.py for a few seconds (2498 geometries passed in a 2-minute run): odd sizes, tiles cut by the
    edge, 1-4 clips, T in {3,5,7}; plain / fused 3x3 and the 1x1, repeatability included."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_bf16
    n, worst = stress_bf16.run(seed=7, seconds=8.0, max_iters=60)
    assert n >= 5 and worst < 2.0 ** -7 + 1e-3


@pytest.mark.parametrize("clips,T,H,W", [(2, 7, 20, 36), (1, 5, 32, 64), (3, 3, 17, 33), (1, 7, 8, 32), (2, 7, 70, 100)])
def test_conv1_conv10_bf16_one_launch(clips, T, H, W):
    """MODE 2 of the bf16 3x3 kernel: conv1_i's output identical (bit for bit) to the plain launch, `base` within one
    bf16 ulp of the 1x1 computed from that (already rounded) output with fp32 accumulation."""
    g = torch.Generator().manual_seed(clips * 100 + T + H)
    Fr = clips * T
    x = r16(torch.randn(Fr, H, W, 64, generator=g))
    k1 = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
    b1 = (torch.randn(64, generator=g) * 0.1).numpy()
    k10 = (torch.randn(1, 1, 64 * T, 64, generator=g) * 0.05).numpy()
    b10 = (torch.randn(64, generator=g) * 0.1).numpy()
    xb = x.to(torch.bfloat16).cuda()
    out1, base = ops.conv1_conv10_bf16(xb, k1, b1, k10, b10, T)
    ref1 = ops.conv3x3_bf16(xb, k1, b1, act=True)
    assert torch.equal(out1, ref1)
    o = out1.float().cpu()
    cat = o.reshape(clips, T, H, W, 64).permute(0, 2, 3, 1, 4).reshape(clips, H, W, T * 64)
    close_bf16(base, F.leaky_relu(cat @ r16(torch.from_numpy(k10[0, 0])) + torch.from_numpy(b10), 0.2))
    assert torch.equal(base, ops.conv1_conv10_bf16(xb, k1, b1, k10, b10, T)[1])


def test_dropin_class_on_the_bf16_path(tmp_path):
    """PFNL().test_video_lr with `precision = "bf16"` (or PFNL_PRECISION=bf16) on a tiny PNG sequence: uint8 frames within
    one grey level of the fp32 golden frames (the harness clips and rounds, reference model/pfnl.py:255-257)."""
    from PIL import Image
    from conftest import load_golden
    from model.pfnl import PFNL
    gd = load_golden("harness_5x16x24_nb1")
    seq = tmp_path / "seq0"
    (seq / "blur4").mkdir(parents=True)
    for i, im in enumerate(gd["lr_u8"]):
        Image.fromarray(im).save(seq / "blur4" / f"{i:04d}.png")
    m = PFNL()
    m.num_block = 1
    m.precision = "bf16"
    m.set_weights(synth.synthetic_weights(PFNLGeometry(num_block=1), seed=0))
    m.save_dir = str(tmp_path / "ckpt")
    m.save(None, m.save_dir, 1000)
    m.test_video_lr(str(seq), name="result", part=2)
    outs = sorted((seq / "result").glob("*.png"))
    assert [p.name for p in outs] == [f"{i:04d}.png" for i in range(5)]
    got = np.stack([np.asarray(Image.open(p)) for p in outs])
    diff = np.abs(got.astype(np.int32) - gd["sr_u8"].astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.05


@pytest.mark.parametrize("clips,T,H,W,cout", [(2, 7, 20, 36, 48), (1, 5, 32, 64, 48), (3, 3, 17, 33, 64), (1, 7, 8, 32, 48), (2, 7, 70, 100, 48)])
def test_conv3x3_accum_bf16(clips, T, H, W, cout):
    """convmerge1 in the accumulating mode of the bf16 3x3 kernel (weight pack swapped in LDS between the frames of a chain):
    fp32 accumulation over all T*576 products, fp32 output -> close to an fp32 conv over the same bf16-rounded operands."""
    g = torch.Generator().manual_seed(clips * 100 + T + H)
    x = r16(torch.randn(clips * T, H, W, 64, generator=g))
    k = (torch.randn(3, 3, 64 * T, cout, generator=g) / np.sqrt(576 * T))
    b = torch.randn(cout, generator=g) * 0.1
    out = ops.conv3x3_accum_bf16(x.to(torch.bfloat16).cuda(), k.numpy(), b.numpy(), act=True, frames_per_clip=T).cpu()
    cat = x.reshape(clips, T, H, W, 64).permute(0, 2, 3, 1, 4).reshape(clips, H, W, T * 64)
    ref = F.leaky_relu(F.conv2d(cat.permute(0, 3, 1, 2), r16(k).permute(3, 2, 0, 1), b, padding=1), 0.2).permute(0, 2, 3, 1)
    assert float((out[..., :cout] - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert float(out[..., cout:].abs().max()) == 0.0 if cout < 64 else True
    assert torch.equal(out, ops.conv3x3_accum_bf16(x.to(torch.bfloat16).cuda(), k.numpy(), b.numpy(), act=True, frames_per_clip=T).cpu())


@pytest.mark.parametrize("name", ["ragged_7x20x36_nb2", "x2_5x16x24_nb2"])
def test_forward_bf16_against_committed_golden(name):
    """The bf16 path against committed fixtures (tests/golden/bf16_outputs.npz, written by tools/make_golden.py from the fast
    oracle's bf16 mode in the build container): PSNR > 60 dB, element-wise within 2e-3 on [0,1]-scale outputs."""
    from conftest import geometry_of, load_golden
    gd = load_golden(name)
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_outputs.npz"))[name]
    geom = geometry_of(gd["meta"])
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(synth.synthetic_weights(geom, seed=0))
    eng.set_option("precision", "bf16")
    y = eng.forward(gd["x"])
    assert y.shape == ref.shape
    assert synth.psnr(y, ref) > 60.0 and np.abs(y - ref).max() < 2e-3
    eng.close()


@pytest.mark.parametrize("sel", ["v2", "v3-all", "v3-mode2"])
def test_bf16_3x3_kernel_generations(sel):
    """Two generations of the bf16 3x3 kernel serve modes 0 - 2 (the first one keeps the accumulating mode only since round 5).  Default:
    conv_bf16_v3.hip (the two halves of the workgroup half a tile period apart) for conv1_i + conv10_i and the per-frame half of conv2_i,
    conv_bf16_v2.hip (halo by LDS-DMA, a serial epilogue phase) for the plain mode.  PFNL_BF16_V3 = 0 / 1 / 2 selects the other
    assignments (same arithmetic: the A/B switch and the bit-equality reference of tools/soak_r04.py); the choice is read once per
    process, so each setting runs the 3x3 op tests, the random-geometry stress and the forward tests in a process of its own."""
    import os
    import subprocess
    import sys
    extra = {"v2": {"PFNL_BF16_V3": "0"}, "v3-all": {"PFNL_BF16_V3": "1"}, "v3-mode2": {"PFNL_BF16_V3": "2"}}[sel]
    env = dict(os.environ, **extra)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_bf16.py"), "-x", "-q", "-m", "gpu", "-k",
                        "conv3x3_bf16_plain or conv3x3_bf16_fused or conv1_conv10_bf16 or random_geometries or matches_rounded_oracle"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_round4_schedules_are_deterministic_and_bit_equal_across_generations():
    """tools/soak_r04.py: the kernels whose LDS traffic is ordered by counted `s_waitcnt vmcnt` waits (the third-generation bf16 3x3
    kernel, the software-pipelined non-local kernel) - every forward repeated and compared bit for bit with its first run, and the bf16
    forward bit for bit with the second-generation kernels' output (same arithmetic, other schedule)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_r04.py"), "4"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "soak_r04: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-800:]

