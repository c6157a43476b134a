"""Per-op parity on a real MI355X: each HIP kernel class against the fp64 spec oracle
(oracle/pfnl_spec.py), through the C-ABI op hooks.  Tolerances: f32 MFMA is an exact fmaf chain
(cdna guide §3), so differences are summation-order round-off: |err| <= 2e-6 * sum|a*b| scale."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

torch = pytest.importorskip("torch")

from oracle import pfnl_spec  # noqa: E402
from pfnl_amd import ops  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def test_library_loaded_and_mfma_layout():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    ops.selftest_mfma(0)


@pytest.mark.parametrize("B,H,W,scale", [(1, 5, 7, 4), (2, 16, 24, 4), (1, 9, 6, 2), (1, 1, 1, 4), (1, 2, 33, 4)])
def test_bicubic(B, H, W, scale):
    rng = np.random.default_rng(H * 100 + W)
    x = rng.random((B, H, W, 3), dtype=np.float32)
    got = ops.bicubic(dev(x), scale).cpu().numpy()
    ref = pfnl_spec.resize_bicubic_tf1(x.astype(np.float64), scale)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-6
    assert np.array_equal(got[:, ::scale, ::scale], x)          # t = 0 taps are exactly [0,1,0,0]


def _conv_case(rng, items, fpi, H, W, ks, cout, act, fused):
    cin = 64 * fpi
    x = rng.normal(size=(items * fpi, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(ks, ks, cin, cout)) / np.sqrt(ks * ks * cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32) * 0.1
    xin = x.reshape(items, fpi, H, W, 64).transpose(0, 2, 3, 1, 4).reshape(items, H, W, cin)   # concat over frames
    ref = pfnl_spec.conv2d_same(xin.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
    kw = {}
    if fused:
        div = 7 if items % 7 == 0 else 1
        add = rng.normal(size=(items // div, H, W, 64)).astype(np.float32)
        res = rng.normal(size=(items, H, W, 64)).astype(np.float32)
        ref = ref + np.repeat(add.astype(np.float64), div, axis=0)
        kw = dict(addend=dev(add), add_div=div, resid=dev(res))
    if act:
        ref = pfnl_spec.lrelu(ref)
    if fused:
        ref = ref + res
    got = ops.conv2d(dev(x), k, b, act=act, frames_per_item=fpi, **kw).cpu().numpy()
    return got, ref


@pytest.mark.parametrize("items,fpi,H,W,ks,cout,act,fused", [
    (2, 1, 8, 32, 3, 64, True, False),       # exactly one tile
    (3, 1, 20, 36, 3, 64, True, False),      # ragged tiles in both directions
    (7, 1, 10, 34, 3, 64, True, True),       # conv2 epilogue: + addend(item/7) , lrelu, + resid
    (2, 1, 9, 70, 3, 64, False, False),      # raw (the shared-base half of conv2)
    (2, 7, 16, 40, 1, 64, True, False),      # conv10: 1x1 over 7 concatenated frames
    (1, 7, 12, 33, 3, 48, True, False),      # convmerge1: 3x3 over 448 -> 48
    (1, 5, 6, 8, 1, 64, True, False),        # T = 5
    (1, 1, 1, 1, 3, 64, True, False),        # a single pixel: everything is halo
])
@pytest.mark.parametrize("mt", ["1", "2"])     # rows per wave: the launcher picks by grid size, both are forced here
def test_conv_mfma(items, fpi, H, W, ks, cout, act, fused, mt, monkeypatch):
    monkeypatch.setenv("PFNL_CONV_MT", mt)
    rng = np.random.default_rng(items * 1000 + H * 10 + W + ks)
    got, ref = _conv_case(rng, items, fpi, H, W, ks, cout, act, fused)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    assert err < 5e-6 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("clips,T,H,W", [
    (1, 7, 4, 32),        # one tile, one group
    (2, 7, 20, 36),       # ragged tiles, two clips
    (3, 3, 6, 70),        # T = 3
    (1, 5, 2, 2),         # a single 2x2 tile: everything is halo
    (4, 7, 64, 64),       # several groups per workgroup
])
def test_conv2_grouped(clips, T, H, W):
    """conv2_i as written in the reference: conv over concat([base, frame]) (model/pfnl.py:69-71)."""
    rng = np.random.default_rng(clips * 100 + T * 10 + H + W)
    x = rng.normal(size=(clips * T, H, W, 64)).astype(np.float32)
    base = rng.normal(size=(clips, H, W, 64)).astype(np.float32)
    res = rng.normal(size=(clips * T, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 128, 64)) / 34).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    cat = np.concatenate([np.repeat(base, T, axis=0), x], axis=-1)
    ref = res + pfnl_spec.lrelu(pfnl_spec.conv2d_same(cat.astype(np.float64), k.astype(np.float64), b.astype(np.float64)))
    got = ops.conv2_grouped(dev(x), dev(base), k, b, dev(res), T).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("clips,T,H,W,cout", [
    (1, 7, 12, 34, 48),     # convmerge1: 448 -> 48, ragged tiles
    (2, 5, 4, 32, 48),      # T = 5, exactly one tile per clip
    (3, 3, 6, 70, 64),      # T = 3, full 64 outputs
    (4, 7, 32, 64, 48),     # several groups per workgroup
    (1, 1, 2, 2, 48),       # degenerate: a single frame, a single 2x2 tile
    (2, 7, 40, 96, 48),     # several chains per workgroup on the persistent kernels
])
@pytest.mark.parametrize("variant", ["winograd", "split16"])   # f32 Winograd / f16 MFMA with exactly split fp32 operands
def test_conv3x3_accum(clips, T, H, W, cout, variant):
    rng = np.random.default_rng(clips * 100 + T * 10 + H + W)
    x = rng.normal(size=(clips * T, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64 * T, cout)) / np.sqrt(9 * 64 * T)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    xin = x.reshape(clips, T, H, W, 64).transpose(0, 2, 3, 1, 4).reshape(clips, H, W, 64 * T)
    ref = pfnl_spec.lrelu(pfnl_spec.conv2d_same(xin.astype(np.float64), k.astype(np.float64), b.astype(np.float64)))
    got = ops.conv3x3_accum(dev(x), k, b, act=True, frames_per_clip=T, variant=variant).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    if variant == "split16" and (H * W) % 2:
        return
    if variant == "split16":                                          # odd sizes too: the split-f16 kernel has no parity restriction
        xo = x[:, :H - 1, :W - 1].copy()
        ref = pfnl_spec.lrelu(pfnl_spec.conv2d_same(xo.reshape(clips, T, H - 1, W - 1, 64).transpose(0, 2, 3, 1, 4).reshape(
            clips, H - 1, W - 1, 64 * T).astype(np.float64), k.astype(np.float64), b.astype(np.float64)))
        got = ops.conv3x3_accum(dev(xo), k, b, act=True, frames_per_clip=T, variant=variant).cpu().numpy()
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("items,fpi,H,W,act", [
    (2, 7, 16, 40, True),        # conv10 at T = 7
    (1, 5, 6, 8, True),          # T = 5, 48 pixels: one full + one ragged wave
    (3, 3, 5, 7, False),         # T = 3, 35 pixels, no activation
    (1, 7, 1, 1, True),          # a single pixel
    (4, 7, 32, 36, True),        # several workgroups per clip
])
@pytest.mark.parametrize("variant", ["stream", "split16"])   # f32 MFMA / f16 MFMA with exactly split fp32 operands (the default)
def test_conv1x1_stream(items, fpi, H, W, act, variant):
    rng = np.random.default_rng(items * 100 + fpi * 10 + H + W)
    x = rng.normal(size=(items * fpi, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(1, 1, 64 * fpi, 64)) / np.sqrt(64 * fpi)).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32)
    xin = x.reshape(items, fpi, H, W, 64).transpose(0, 2, 3, 1, 4).reshape(items, H, W, 64 * fpi)
    ref = pfnl_spec.conv2d_same(xin.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
    if act:
        ref = pfnl_spec.lrelu(ref)
    got = ops.conv1x1_stream(dev(x), k, b, act=act, frames_per_item=fpi, variant=variant).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 5e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("items,H,W,act,fused", [
    (2, 4, 32, True, False),        # exactly one workgroup tile
    (3, 20, 36, True, False),       # ragged in both directions
    (7, 10, 34, True, True),        # conv2 epilogue: + addend(item/7), lrelu, + resid
    (2, 6, 70, False, False),       # raw (the shared-base half of conv2)
    (1, 2, 2, True, False),         # a single 2x2 tile: everything is halo
    (4, 64, 64, True, True),
])
@pytest.mark.parametrize("variant", ["winograd", "winograd_ws"])
def test_conv3x3_winograd(items, H, W, act, fused, variant):
    rng = np.random.default_rng(items * 1000 + H * 10 + W)
    x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64, 64)) / np.sqrt(9 * 64)).astype(np.float32)
    b = rng.normal(size=64).astype(np.float32) * 0.1
    ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
    kw = {}
    if fused:
        div = 7 if items % 7 == 0 else 1
        add = rng.normal(size=(items // div, H, W, 64)).astype(np.float32)
        res = rng.normal(size=(items, H, W, 64)).astype(np.float32)
        ref = ref + np.repeat(add.astype(np.float64), div, axis=0)
        kw = dict(addend=dev(add), add_div=div, resid=dev(res))
    if act:
        ref = pfnl_spec.lrelu(ref)
    if fused:
        ref = ref + res
    got = ops.conv3x3_winograd(dev(x), k, b, act=act, variant=variant, **kw).cpu().numpy()
    direct = ops.conv2d(dev(x), k, b, act=act, **kw).cpu().numpy()
    err = np.abs(got - ref).max()
    assert err < 2e-5 * max(1.0, np.abs(ref).max()), err          # Winograd: a few ulp more than direct
    assert np.abs(got - direct).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_conv_delta_kernel_shift_is_exact():
    rng = np.random.default_rng(5)
    x = rng.random((1, 12, 40, 64), dtype=np.float32)
    k = np.zeros((3, 3, 64, 64), np.float32)
    for c in range(64):
        k[0, 2, c, (c * 7 + 3) % 64] = 1.0          # tap (dy=-1,dx=+1) + a channel permutation
    got = ops.conv2d(dev(x), k, None, act=False).cpu().numpy()
    ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), None)
    assert np.array_equal(got, ref.astype(np.float32))            # pure data movement: bit exact


@pytest.mark.parametrize("B,T,H,W", [(1, 7, 8, 8), (2, 7, 20, 36), (1, 5, 16, 24), (1, 3, 12, 40), (1, 7, 2, 2),
                                     (1, 7, 32, 32), (1, 7, 48, 48), (2, 5, 64, 48), (1, 7, 62, 70)])   # last three: key-split path
@pytest.mark.parametrize("kernel", ["fp32", "split16"])     # f32 MFMA (nonlocal.hip) / f16 MFMA with exactly split operands (nonlocal_f16.hip)
def test_nonlocal_residual(B, T, H, W, kernel):
    rng = np.random.default_rng(B + T + H + W)
    C = 12 * T
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    wg = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    ww = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    bg = rng.normal(size=C).astype(np.float32) * 0.1
    bw = rng.normal(size=C).astype(np.float32) * 0.1
    got = ops.nonlocal_residual(dev(x), wg, bg, ww, bw, precision=kernel).cpu().numpy()
    x64 = x.astype(np.float64)
    stack = np.concatenate([x64[:, t] for t in range(T)], -1)
    z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64),
                                 ww.astype(np.float64), bw.astype(np.float64))
    ref = stack + pfnl_spec.depth_to_space2(z)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()


@pytest.mark.parametrize("N", [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 193, 321])
@pytest.mark.parametrize("kernel", ["split16", "f16"])
def test_nonlocal_key_counts_around_tile_edges(N, kernel):
    """nl_attn_f16_sw_kernel (round 4) masks the keys past N through DATA - key rows past N carry -65504 in the pad channel, the query
    operand 1024 - and pipelines over 32-key halves of 64-key tiles in a ring of 3 / 5 LDS slots: every N around the half, tile and
    ring boundaries, a half that is entirely masked (N = 1, 31, 32, 65 ...), against the fp64 spec."""
    T, H, W = 3, 2, 2 * N
    rng = np.random.default_rng(N)
    C = 12 * T
    x = rng.random((1, T, H, W, 3), dtype=np.float32)
    wg = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    ww = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    bg = rng.normal(size=C).astype(np.float32) * 0.1
    bw = rng.normal(size=C).astype(np.float32) * 0.1
    got = ops.nonlocal_residual(dev(x), wg, bg, ww, bw, precision=kernel).cpu().numpy()
    x64 = x.astype(np.float64)
    stack = np.concatenate([x64[:, t] for t in range(T)], -1)
    z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64),
                                 ww.astype(np.float64), bw.astype(np.float64))
    ref = stack + pfnl_spec.depth_to_space2(z)
    tol = 2e-5 if kernel == "split16" else 2e-3                     # (f16: binary16 operands, the non-local block of precision=bf16)
    assert np.abs(got - ref).max() < tol, np.abs(got - ref).max()


def test_nonlocal_f16_random_geometries_short():
    """tools/stress_nl.py (3 900 geometries clean when the software-pipelined kernel was committed): a short run of it - random B, T, H, W
    against the fp64 spec in both operand forms."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_nl
    n, worst = stress_nl.run(seed=11, seconds=8.0, max_iters=60)
    assert n >= 10 and worst["split16"] < 2e-5 and worst["f16"] < 2e-3, (n, worst)


def test_nonlocal_f16_clip_chunks():
    """nl_attn_f16_sw_kernel reaches its packed operands through ONE buffer resource with 32-bit offsets: a batch whose operands exceed
    ~2 GB runs in chunks of whole clips through the same scratch (launch_nl_attn_f16; the first-generation kernel that used to serve such
    batches left the library in round 5).  The limit is a testing hook (PFNL_NL_CHUNK_BYTES, read once per process): the non-local tests
    again with a limit of two clips' worth at 16x24 - every multi-clip case runs chunked - in a process of their own."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    npad = (8 * 12 + 31) // 32 * 32 + 64                           # N = 96 keys at 16x24
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_ops.py"), os.path.join(here, "test_gpu_bf16.py"),
                        os.path.join(here, "test_gpu_forward.py"), "-x", "-q", "-m", "gpu",
                        "-k", "(nonlocal and not clip_chunks) or forward_matches_golden or full_size_all_clips"],
                       capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, PFNL_NL_CHUNK_BYTES=str(2 * 2 * 4 * 96 * npad)), cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("kernel", ["fp32", "split16"])
def test_nonlocal_constant_and_peaked_inputs(kernel):
    """Known answers: constant frames -> uniform affinity; a very bright pixel block -> the running-max
    rescale branch of the streaming softmax is exercised (logits ~ 84 vs ~ 2)."""
    T, H, W = 7, 16, 16
    C = 12 * T
    rng = np.random.default_rng(0)
    wg = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    ww = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
    bg = rng.normal(size=C).astype(np.float32) * 0.1
    bw = rng.normal(size=C).astype(np.float32) * 0.1
    x = np.full((1, T, H, W, 3), 0.25, np.float32)
    got = ops.nonlocal_residual(dev(x), wg, bg, ww, bw, precision=kernel).cpu().numpy()
    g = np.full(C, 0.25) @ wg[0, 0].astype(np.float64) + bg
    zc = g @ ww[0, 0].astype(np.float64) + bw
    ref = 0.25 + pfnl_spec.depth_to_space2(np.broadcast_to(zc, (1, H // 2, W // 2, C)).copy())
    assert np.abs(got - ref).max() < 1e-5
    x = (rng.random((1, T, H, W, 3)) * 0.15).astype(np.float32)
    x[:, :, 10:14, 4:8] = 0.97 + 0.03 * rng.random((1, T, 4, 4, 3)).astype(np.float32)   # late, dominant keys
    got = ops.nonlocal_residual(dev(x), wg, bg, ww, bw, precision=kernel).cpu().numpy()
    x64 = x.astype(np.float64)
    stack = np.concatenate([x64[:, t] for t in range(T)], -1)
    z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64),
                                 ww.astype(np.float64), bw.astype(np.float64), stabilise=True)
    ref = stack + pfnl_spec.depth_to_space2(z)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 5e-5


@pytest.mark.parametrize("F,H,W,scale", [(2, 24, 32, 4), (1, 37, 29, 4), (3, 16, 16, 2), (1, 7, 7, 4)])
def test_blur_decimate(F, H, W, scale):
    from pfnl_amd import synth
    rng = np.random.default_rng(F + H + W)
    hr = rng.random((F, H, W, 3), dtype=np.float32)
    got = ops.blur_decimate(dev(hr), scale).cpu().numpy()
    ref = synth.blur_decimate(hr.astype(np.float64), scale)        # numpy restatement of utils.py:169-192
    assert got.shape == ref.shape == (F, -(-H // scale), -(-W // scale), 3)
    assert np.abs(got - ref).max() < 2e-6


def test_ws_modes_random_geometries():
    """Short run of tools/stress_ws.py: modes 0 / 2 / 3 of the persistent Winograd kernel against the per-tile and
    direct kernels over random (T, clips, H, W), and bit-exact repeatability of every call."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_ws", os.path.join(os.path.dirname(HERE), "tools", "stress_ws.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, worst = mod.run(seed=7, seconds=60.0, max_iters=25)
    assert n == 25 and worst < 3e-5


# ---- round 2: op hooks for the head and the tail, the embedded-Gaussian block, the harness helpers ------------------

@pytest.mark.parametrize("B,T,H,W", [(1, 7, 16, 24), (2, 3, 10, 38), (1, 5, 34, 18), (1, 7, 2, 2)])
def test_conv0_op(B, T, H, W):
    """conv0 (reference model/pfnl.py:48,61-62): lrelu(conv5x5 'same' 3->64 + b) per frame vs the fp64 spec."""
    rng = np.random.default_rng(B * 1000 + H * 10 + W)
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    k = (rng.normal(size=(5, 5, 3, 64)) / np.sqrt(75.0)).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    got = ops.conv0(dev(x), k, b).cpu().numpy()
    ref = pfnl_spec.lrelu(pfnl_spec.conv2d_same(x.reshape(B * T, H, W, 3).astype(np.float64), k.astype(np.float64), b.astype(np.float64)))
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())
    delta = np.zeros((5, 5, 3, 64), np.float32)                       # pure data movement: a shifted delta kernel
    delta[1, 3, 2, 7] = 1.0
    got = ops.conv0(dev(x), delta, None).cpu().numpy()
    xp = np.pad(x.reshape(B * T, H, W, 3), ((0, 0), (2, 2), (2, 2), (0, 0)))
    want = xp[:, 1:1 + H, 3:3 + W, 2]
    # the f16-pipe kernel carries an fp32 value as f16(x) + f16((x - hi) 2^11) 2^-11: 22 of its 24 significant bits
    assert np.abs(got[..., 7] - want).max() <= 2.0 ** -22            # (|x| < 1: absolute = relative bound)
    assert not got[..., :7].any() and not got[..., 8:].any()


@pytest.mark.parametrize("B,T,H,W,scale", [(1, 7, 8, 12, 4), (2, 5, 6, 20, 2), (1, 3, 17, 5, 4), (1, 7, 1, 1, 4)])
def test_tail_op(B, T, H, W, scale):
    """d2s -> convmerge2 (no activation) -> d2s (4x) -> + legacy bicubic of the centre frame (reference model/pfnl.py:53,63,
    76-80) vs the fp64 spec, op for op."""
    rng = np.random.default_rng(H * 31 + W)
    merge = rng.normal(size=(B, H, W, 48)).astype(np.float32)
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    co = 12 if scale == 4 else 3
    k = (rng.normal(size=(3, 3, 12, co)) / np.sqrt(108.0)).astype(np.float32)
    b = (rng.normal(size=co) * 0.1).astype(np.float32)
    got = ops.tail(dev(merge), dev(x), k, b, scale).cpu().numpy()
    o = pfnl_spec.conv2d_same(pfnl_spec.depth_to_space2(merge.astype(np.float64)), k.astype(np.float64), b.astype(np.float64))
    if scale == 4:
        o = pfnl_spec.depth_to_space2(o)
    ref = (o + pfnl_spec.resize_bicubic_tf1(x[:, T // 2].astype(np.float64), scale))[:, None]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 5e-6


@pytest.mark.parametrize("B,T,H,W", [(1, 7, 16, 16), (2, 3, 6, 10), (1, 5, 12, 22)])
def test_nonlocal_embedded_gaussian(B, T, H, W):
    """reference utils.py:18-71 with nltype=0 (theta / phi 1x1 projections in front of the affinity) + the residual of
    model/pfnl.py:55-60, against the fp64 spec written as the reference (two convs, full N x N affinity)."""
    rng = np.random.default_rng(T * 7 + H)
    C = 12 * T
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    mk = lambda sc: (rng.normal(size=(1, 1, C, C)) * sc).astype(np.float32)      # noqa: E731
    mb = lambda: (rng.normal(size=C) * 0.05).astype(np.float32)                  # noqa: E731
    wg, ww, wt, wp = mk(0.1), mk(0.1), mk(0.15), mk(0.15)
    bg, bw, bt, bp = mb(), mb(), mb(), mb()
    got = ops.nonlocal_embedded(dev(x), wg, bg, ww, bw, wt, bt, wp, bp).cpu().numpy()
    stack = np.concatenate([x[:, t] for t in range(T)], -1).astype(np.float64)
    x1 = pfnl_spec.space_to_depth2(stack)
    f = lambda a: a.astype(np.float64)                                           # noqa: E731
    z = pfnl_spec.nonlocal_block(x1, f(wg), f(bg), f(ww), f(bw), theta=(f(wt), f(bt)), phi=(f(wp), f(bp)))
    ref = stack + pfnl_spec.depth_to_space2(z)
    assert np.abs(got - ref).max() < 2e-5
    plain = ops.nonlocal_residual(dev(x), wg, bg, ww, bw).cpu().numpy()
    assert np.abs(plain - ref).max() > 1e-4                                      # theta/phi matter


@pytest.mark.parametrize("B,T,H,W,nltype,sub", [
    (1, 7, 16, 16, 2, 1), (2, 3, 6, 10, 2, 1), (1, 5, 12, 22, 2, 2),          # dot product (relu / rowsum), plain and pooled
    (1, 7, 16, 24, 1, 2), (1, 7, 18, 14, 1, 3), (2, 5, 20, 12, 1, 2),        # PFNL's Gaussian with pooled keys (ragged: 9x7 grid / 3)
    (1, 7, 16, 16, 0, 2), (1, 3, 22, 10, 0, 3), (1, 7, 40, 72, 1, 2),        # embedded Gaussian pooled; a key-split geometry
    (1, 7, 40, 72, 2, 1), (1, 7, 4, 4, 1, 2), (1, 5, 12, 8, 2, 3)])          # ... for the dot product; a single pooled key; two
def test_nonlocal_block_general(B, T, H, W, nltype, sub):
    """utils.NonLocalBlock(input_x, C, sub_sample, nltype) as the reference writes it (utils.py:18-71: g / phi / theta convolutions,
    average pooling of g and phi, full N x Nk affinity, exp- or relu-normalisation) + the wrapper of model/pfnl.py:55-60, in fp64,
    against pfnl_op_nonlocal_block - which folds the convolutions, pools X instead of g and phi, and never materialises the affinity."""
    rng = np.random.default_rng(T * 7 + H + 100 * nltype + sub)
    C = 12 * T
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    mk = lambda sc: (rng.normal(size=(1, 1, C, C)) * sc).astype(np.float32)      # noqa: E731
    mb = lambda sc=0.05: (rng.normal(size=C) * sc).astype(np.float32)            # noqa: E731
    wg, ww, wt, wp = mk(0.1), mk(0.1), mk(0.15), mk(0.15)
    bg, bw, bt, bp = mb(), mb(), mb(0.3), mb(0.3)                                # (b_phi matters for nltype 2 only: not small)
    th, ph = ((wt, bt), (wp, bp)) if nltype != 1 else (None, None)
    got = ops.nonlocal_block(dev(x), wg, bg, ww, bw, theta=th, phi=ph, nltype=nltype, sub_sample=sub).cpu().numpy()
    stack = np.concatenate([x[:, t] for t in range(T)], -1).astype(np.float64)
    x1 = pfnl_spec.space_to_depth2(stack)
    f = lambda a: a.astype(np.float64)                                           # noqa: E731
    z = pfnl_spec.nonlocal_block(x1, f(wg), f(bg), f(ww), f(bw), theta=None if th is None else (f(wt), f(bt)),
                                 phi=None if ph is None else (f(wp), f(bp)), nltype=nltype, sub_sample=sub)
    ref = stack + pfnl_spec.depth_to_space2(z)
    ok = np.isfinite(ref)                                                        # (nltype 2: a query without a positive affinity is 0 / 0 in
    assert got.shape == ref.shape and np.array_equal(np.isfinite(got), ok)       #  the reference: NaN on both sides, utils.py:61-63)
    assert ok.mean() > 0.5 and np.abs(got[ok] - ref[ok]).max() < 2e-5, np.abs(got[ok] - ref[ok]).max()
    if nltype == 1 and sub == 1:
        return
    plain = ops.nonlocal_residual(dev(x), wg, bg, ww, bw, precision="f32").cpu().numpy()
    assert np.abs(plain[ok] - ref[ok]).max() > 1e-5                              # the options change the result
    if nltype == 2:                                                              # ... and so does the per-query constant theta . b_phi
        z0 = pfnl_spec.nonlocal_block(x1, f(wg), f(bg), f(ww), f(bw), theta=(f(wt), f(bt)), phi=(f(wp), 0 * f(bp)), nltype=2, sub_sample=sub)
        assert np.nanmax(np.abs(z0 - z)) > 1e-5


def test_nonlocal_block_rejects():
    x = dev(np.zeros((1, 7, 8, 8, 3), np.float32))
    C = 84
    w, b = np.zeros((C, C), np.float32), np.zeros(C, np.float32)
    with pytest.raises(ValueError):
        ops.nonlocal_block(x, w, b, w, b, nltype=3)
    with pytest.raises(ValueError):
        ops.nonlocal_block(x, w, b, w, b, nltype=2)                              # no projections
    with pytest.raises(Exception):
        ops.nonlocal_block(x, w, b, w, b, nltype=1, sub_sample=5)                # 4x4 grid / 5: no key left


def test_gather_windows_and_quantise():
    """reference model/pfnl.py:238-242 (clamped windows) and :254-257 (uint8 quantisation) on the device: bit-exact
    against the host restatements."""
    from pfnl_amd import model as M
    rng = np.random.default_rng(9)
    for F, T, H, W in ((5, 7, 4, 6), (1, 7, 2, 2), (9, 3, 6, 10), (12, 5, 8, 2)):
        fr = rng.random((F, H, W, 3), dtype=np.float32)
        want = M.sliding_windows(fr, T)
        for first, count in ((0, F), (F // 2, F - F // 2), (F - 1, 1)):
            got = ops.gather_windows(dev(fr), first, count, T).cpu().numpy()
            assert np.array_equal(got, want[first:first + count])
    sr = (rng.random((2, 1, 8, 12, 3), dtype=np.float32) * 1.4 - 0.2)
    sr.ravel()[:8] = [0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255, 1.0, 0.0, -1.0, 2.0]      # ties, ends, out of range
    assert np.array_equal(ops.quantise_u8(dev(sr)).cpu().numpy(), M.quantise(sr))


@pytest.mark.parametrize("items,H,W,fused,act", [(1, 8, 32, False, True), (7, 10, 38, True, True), (3, 5, 7, False, False), (1, 1, 1, False, True),
                                                  (7, 33, 70, True, True), (2, 64, 96, False, True), (28, 24, 40, True, True),
                                                  (1, 9, 130, False, True), (21, 16, 32, True, False)])
def test_conv3x3_split16(items, H, W, fused, act):
    """The fp32 3x3 64->64 convolution on the f16 matrix pipe with exactly split operands (conv_split16.hip; reference
    model/pfnl.py:49-51 at :66-71): every product carries >= 22 mantissa bits, accumulation is fp32 - the error against the
    fp64 spec must not exceed the direct f32-MFMA kernel's (an fp32 FMA chain).  Odd sizes, ragged tiles, chains of 7."""
    rng = np.random.default_rng(items * 1000 + H * 10 + W)
    x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
    kw = {}
    if fused:
        div = 7 if items % 7 == 0 else 1
        add = rng.normal(size=(items // div, H, W, 64)).astype(np.float32)
        res = rng.normal(size=(items, H, W, 64)).astype(np.float32)
        ref = ref + np.repeat(add.astype(np.float64), div, axis=0)
        kw = dict(addend=dev(add), add_div=div, resid=dev(res))
    if act:
        ref = pfnl_spec.lrelu(ref)
    if fused:
        ref = ref + res
    got = ops.conv3x3_winograd(dev(x), k, b, act=act, variant="split16", **kw).cpu().numpy()
    direct = ops.conv2d(dev(x), k, b, act=act, **kw).cpu().numpy()
    e_s, e_d = np.abs(got - ref).max(), np.abs(direct - ref).max()
    assert e_s < 4e-6 * max(1.0, np.abs(ref).max()), (e_s, e_d)
    assert e_s <= 1.5 * e_d + 1e-7, (e_s, e_d)                    # as good as the fp32 FMA chain of the direct kernel


@pytest.mark.parametrize("items,H,W,fused,act", [(1, 8, 32, False, True), (7, 10, 38, True, True), (3, 5, 7, False, False), (1, 1, 1, False, True),
                                                  (7, 33, 70, True, True), (2, 64, 96, False, True), (28, 24, 40, True, True),
                                                  (1, 9, 130, False, True), (21, 16, 32, True, False), (4, 128, 128, False, False)])
def test_conv3x3_split16_sf_input(items, H, W, fused, act):
    """conv3x3_sf_kernel (conv_sf.hip; both halves of conv2_i, reference model/pfnl.py:51,69-71): the input in the split format
    comes in by LDS-DMA (source-side swizzle, zero-filled borders), the epilogue runs from registers.  Same arithmetic as
    conv3x3_split16_kernel: the error against the fp64 spec is bounded by the direct f32-MFMA kernel's; and, the operands being
    the same binary16 pairs, it agrees with that kernel to summation order."""
    rng = np.random.default_rng(items * 1000 + H * 10 + W)
    x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
    kw = {}
    if fused:
        div = 7 if items % 7 == 0 else 1
        add = rng.normal(size=(items // div, H, W, 64)).astype(np.float32)
        res = rng.normal(size=(items, H, W, 64)).astype(np.float32)
        ref = ref + np.repeat(add.astype(np.float64), div, axis=0)
        kw = dict(addend=dev(add), add_div=div, resid=dev(res))
    if act:
        ref = pfnl_spec.lrelu(ref)
    if fused:
        ref = ref + res
    got = ops.conv3x3_winograd(dev(x), k, b, act=act, variant="split16_sf_in", **kw).cpu().numpy()
    old = ops.conv3x3_winograd(dev(x), k, b, act=act, variant="split16", **kw).cpu().numpy()
    direct = ops.conv2d(dev(x), k, b, act=act, **kw).cpu().numpy()
    e_s, e_d, e_o = np.abs(got - ref).max(), np.abs(direct - ref).max(), np.abs(got - old).max()
    print(f"sf_in {items}x{H}x{W} fused={fused}: err {e_s:.3g} (direct f32 {e_d:.3g}), vs split16 kernel {e_o:.3g}")
    assert e_s < 4e-6 * max(1.0, np.abs(ref).max()), (e_s, e_d)
    assert e_s <= 1.5 * e_d + 1e-7, (e_s, e_d)
    assert e_o < 2e-6 * max(1.0, np.abs(ref).max()), e_o


@pytest.mark.parametrize("T,clips,H,W", [(7, 1, 8, 32), (7, 2, 10, 38), (5, 1, 33, 70), (3, 3, 16, 24), (7, 4, 128, 128), (7, 1, 1, 1), (7, 3, 9, 130)])
def test_conv2_chain_sf(T, clips, H, W):
    """conv3x3_sf_chain_kernel: the WHOLE of conv2_i (reference model/pfnl.py:69-71: lrelu(conv3x3(concat([base, inp1_t]), 128 -> 64)
    + bias) + inp0_t) in one launch, the shared half held in registers as the initial C of the T frame tiles, weights and halos
    by LDS-DMA.  Against the fp64 spec of the concat form, and against the two-launch form (same operands: summation order only)."""
    rng = np.random.default_rng(T * 1000 + H * 10 + W)
    F = clips * T
    x = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    base = rng.normal(size=(clips, H, W, 64)).astype(np.float32)
    res = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    k2 = (rng.normal(size=(3, 3, 128, 64)) / 34.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    cat = np.concatenate([np.repeat(base.astype(np.float64), T, axis=0), x.astype(np.float64)], axis=-1)
    ref = pfnl_spec.lrelu(pfnl_spec.conv2d_same(cat, k2.astype(np.float64), b.astype(np.float64))) + res
    got = ops.conv3x3_winograd(dev(x), k2, b, act=True, addend=dev(base), add_div=T, resid=dev(res), variant="split16_sf_chain").cpu().numpy()
    pb = ops.conv3x3_winograd(dev(base), np.ascontiguousarray(k2[:, :, :64]), None, act=False, variant="split16_sf_in")
    two = ops.conv3x3_winograd(dev(x), np.ascontiguousarray(k2[:, :, 64:]), b, act=True, addend=pb, add_div=T, resid=dev(res), variant="split16_sf_in").cpu().numpy()
    e, e2 = np.abs(got - ref).max(), np.abs(got - two).max()
    print(f"conv2 chain T{T} {clips}x{H}x{W}: err {e:.3g}, vs the two-launch form {e2:.3g}")
    assert e < 4e-6 * max(1.0, np.abs(ref).max()), e
    assert e2 < 2e-6 * max(1.0, np.abs(ref).max()), e2


@pytest.mark.parametrize("items,H,W,act", [(1, 8, 32, True), (3, 5, 7, False), (1, 1, 1, True), (2, 64, 96, True), (7, 33, 70, True), (1, 9, 130, True)])
def test_conv3x3_split16_sf_output(items, H, W, act):
    """conv1_i writing the split format (conv3x3_split16_kernel<0, OSF>): hi + lo' 2^-11 of what it writes is the fp32 kernel's
    output to 2^-22 relative (the split keeps 22 mantissa bits) - and hi is exactly f16(value), lo' exactly f16((value - hi) 2^11),
    which the consumers' parity tests cover by construction (same operands as the in-kernel split)."""
    rng = np.random.default_rng(items * 77 + H * 10 + W)
    x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    got = ops.conv3x3_winograd(dev(x), k, b, act=act, variant="split16_sf_out").cpu().numpy()
    old = ops.conv3x3_winograd(dev(x), k, b, act=act, variant="split16").cpu().numpy()
    rel = np.abs(got - old) / np.maximum(np.abs(old), 2.0 ** -14)
    print(f"sf_out {items}x{H}x{W}: max rel {rel.max():.3g}")
    assert rel.max() <= 2.0 ** -21, rel.max()


@pytest.mark.parametrize("T,items,H,W", [(7, 2, 16, 32), (5, 1, 6, 10), (3, 3, 9, 7), (7, 4, 64, 64)])
@pytest.mark.parametrize("io", ["10", "01", "11"])
def test_conv1x1_split16_sf(T, items, H, W, io):
    """conv10_i with its input and / or output in the split format: against the fp64 spec at the tolerance of the fp32-interface
    kernel, and against that kernel."""
    rng = np.random.default_rng(T * 100 + H)
    x = rng.normal(size=(items * T, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(1, 1, 64 * T, 64)) / np.sqrt(64 * T)).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    xc = x.reshape(items, T, H, W, 64).transpose(0, 2, 3, 1, 4).reshape(items, H, W, T * 64)
    ref = pfnl_spec.lrelu(pfnl_spec.conv2d_same(xc.astype(np.float64), k.astype(np.float64), b.astype(np.float64)))
    got = ops.conv1x1_stream(dev(x), k, b, frames_per_item=T, act=True, variant="split16_sf:" + io).cpu().numpy()
    old = ops.conv1x1_stream(dev(x), k, b, frames_per_item=T, act=True, variant="split16").cpu().numpy()
    e, e_o = np.abs(got - ref).max(), np.abs(got - old).max()
    print(f"conv1x1 sf {io} T{T} {items}x{H}x{W}: err {e:.3g}, vs fp32-interface kernel {e_o:.3g}")
    assert e < 4e-6 * max(1.0, np.abs(ref).max()), e
    assert e_o < 2e-6 * max(1.0, np.abs(ref).max()), e_o


@pytest.mark.parametrize("T,clips,H,W", [(7, 1, 8, 32), (7, 2, 16, 64), (5, 1, 9, 38), (3, 3, 5, 7), (7, 1, 1, 1), (7, 1, 33, 70),
                                          (7, 4, 128, 128), (5, 2, 64, 96), (1, 2, 24, 40), (7, 40, 8, 32)])
def test_conv1_conv10_fused_split16(T, clips, H, W):
    """conv1_i + conv10_i as ONE launch (conv3x3_c1c10_kernel; reference model/pfnl.py:66-68): inp1 and base against the fp64 spec
    (two convolutions, written as the reference) and against the two-launch form it replaces; ragged tiles, one tile per chain, more
    chains than workgroups (40 clips x 1 tile), the configs[1] geometry."""
    rng = np.random.default_rng(T * 1000 + H * 10 + W + clips)
    F = clips * T
    x = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    k1 = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b1 = (rng.normal(size=64) * 0.1).astype(np.float32)
    k10 = (rng.normal(size=(1, 1, 64 * T, 64)) / np.sqrt(64 * T)).astype(np.float32)
    b10 = (rng.normal(size=64) * 0.1).astype(np.float32)
    got1, gotb = (t.cpu().numpy() for t in ops.conv1_conv10_split16(dev(x), k1, b1, k10, b10, T))
    two1 = ops.conv3x3_winograd(dev(x), k1, b1, act=True, variant="split16_sf_out")
    twob = ops.conv1x1_stream(two1, k10, b10, frames_per_item=T, act=True, variant="split16_sf:01").cpu().numpy()
    big = F * H * W > 200000                                                        # fp64 spec on a subset of the clips only
    nc = 1 if big else clips
    ref1 = pfnl_spec.lrelu(pfnl_spec.conv2d_same(x[:nc * T].astype(np.float64), k1.astype(np.float64), b1.astype(np.float64)))
    cat = ref1.reshape(nc, T, H, W, 64).transpose(0, 2, 3, 1, 4).reshape(nc, H, W, T * 64)
    refb = pfnl_spec.lrelu(pfnl_spec.conv2d_same(cat, k10.astype(np.float64), b10.astype(np.float64)))
    e1, eb = np.abs(got1[:nc * T] - ref1).max(), np.abs(gotb[:nc] - refb).max()
    d1, db = np.abs(got1 - two1.cpu().numpy()).max(), np.abs(gotb - twob).max()
    print(f"conv1+conv10 fused T{T} {clips}x{H}x{W}: inp1 err {e1:.3g} (two-launch {d1:.3g}), base err {eb:.3g} (two-launch {db:.3g})")
    assert e1 < 4e-6 * max(1.0, np.abs(ref1).max()) and eb < 4e-6 * max(1.0, np.abs(refb).max()), (e1, eb)
    assert d1 <= 2.0 ** -20 * max(1.0, np.abs(ref1).max()), d1                      # the same MFMAs in the same order (the fold of the cross terms
                                                                                     # may contract differently): one step of the 22-bit split
    assert db < 2e-6 * max(1.0, np.abs(refb).max()), db


@pytest.mark.parametrize("T,clips,H,W", [(7, 1, 8, 32), (7, 2, 16, 64), (5, 1, 9, 38), (3, 3, 5, 7), (7, 1, 1, 1), (7, 1, 33, 70),
                                          (7, 4, 128, 128), (5, 2, 64, 96), (1, 2, 24, 40), (7, 40, 8, 32), (7, 2, 50, 34), (7, 300, 8, 16)])
def test_conv1_conv10_fused_split16_sf0_is_bit_identical(T, clips, H, W):
    """Round 6 (option split16_sf0; reference model/pfnl.py:66-68): conv3x3_c1c10_kernel<true> takes its halo by LDS-DMA from the
    split-format copy of inp0 instead of splitting fp32 on the VALU when it commits a halo tile.  The operands are the same binary16
    pairs in the same order, so inp1 and base must be BIT-IDENTICAL to the fp32-input launch - on ragged tiles, single pixels, image
    borders inside a tile (out-of-range DMA offsets = zeros), more chains than workgroups and the configs[1] geometry.  Repeated: the DMA
    lands behind a fence load, a late piece would be a wrong tile."""
    rng = np.random.default_rng(T * 1000 + H * 10 + W + clips)
    F = clips * T
    x = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    x[0, 0, 0, :8] = [0.0, -0.0, 1e-30, 65000.0, -3.1e4, 2.0 ** -14, 2.0 ** -25, 1.0 + 2.0 ** -12]   # zero, tiny, near binary16's range ends
    k1 = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b1 = (rng.normal(size=64) * 0.1).astype(np.float32)
    k10 = (rng.normal(size=(1, 1, 64 * T, 64)) / np.sqrt(64 * T)).astype(np.float32)
    b10 = (rng.normal(size=64) * 0.1).astype(np.float32)
    ref1, refb = (t.cpu().numpy() for t in ops.conv1_conv10_split16(dev(x), k1, b1, k10, b10, T))
    for rep in range(3):
        got1, gotb = (t.cpu().numpy() for t in ops.conv1_conv10_split16(dev(x), k1, b1, k10, b10, T, sf0=True))
        assert np.array_equal(got1.view(np.uint32), ref1.view(np.uint32)), (rep, np.abs(got1 - ref1).max())
        assert np.array_equal(gotb.view(np.uint32), refb.view(np.uint32)), (rep, np.abs(gotb - refb).max())


def _sf_split_host(v):
    """The split format of an fp32 array [..., 64] as the kernels build it (conv_split16.h): per pixel [channel half][hi 32 | lo' 32]
    binary16, hi = f16(x) (nearest even), lo' = f16((x - hi) 2^11) - x - hi and the scaling are exact in fp32, one rounding each."""
    hi = v.astype(np.float16)
    lo = ((v - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    hi, lo = hi.reshape(v.shape[:-1] + (2, 32)), lo.reshape(v.shape[:-1] + (2, 32))
    return np.concatenate([hi, lo], axis=-1).reshape(v.shape[:-1] + (128,)).view(np.int16)


@pytest.mark.parametrize("T,clips,H,W", [(7, 1, 8, 32), (7, 2, 10, 38), (5, 1, 33, 70), (3, 3, 16, 24), (7, 4, 128, 128), (7, 1, 1, 1), (7, 3, 9, 130), (7, 40, 8, 32)])
def test_conv2_chain_sf0_writes_the_split_format_copy(T, clips, H, W):
    """Round 6 (option split16_sf0; reference model/pfnl.py:69-71): conv3x3_sf_chain_kernel<true> writes its output - the next block's
    inp0 - a second time in the split format (a lane owns one channel there, an SF chunk is 8 channels of a pixel: lane pairs swap
    binary16 pairs by DPP and store hi / lo' dwords).  The fp32 output must be BIT-IDENTICAL to the kernel without the copy, and the
    copy must be exactly the split of that output (what conv3x3_c1c10_kernel<false> would have built from it in LDS)."""
    rng = np.random.default_rng(T * 1000 + H * 10 + W)
    F = clips * T
    x = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    base = rng.normal(size=(clips, H, W, 64)).astype(np.float32)
    res = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    k2 = (rng.normal(size=(3, 3, 128, 64)) / 34.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    plain = ops.conv3x3_winograd(dev(x), k2, b, act=True, addend=dev(base), add_div=T, resid=dev(res), variant="split16_sf_chain").cpu().numpy()
    out, out_sf = ops.conv2_chain_sf0(dev(x), k2, b, dev(base), dev(res), T)
    out, out_sf = out.cpu().numpy(), out_sf.cpu().numpy()
    assert np.array_equal(out.view(np.uint32), plain.view(np.uint32))
    want = _sf_split_host(out)
    bad = np.argwhere(out_sf != want)
    assert bad.size == 0, (len(bad), bad[:4], out_sf[tuple(bad[0])], want[tuple(bad[0])])


@pytest.mark.parametrize("T,clips,H,W", [(7, 1, 32, 32), (5, 1, 64, 64), (7, 2, 9, 38), (3, 1, 2, 2), (7, 1, 33, 70), (5, 3, 16, 24)])
def test_conv_small_trunk_ops(T, clips, H, W):
    """conv_small.hip (BASELINE.json configs[0] / configs[4] path): conv1_i, conv10_i, the whole of conv2_i over
    concat([base, f]) with residual, and convmerge1 (448 -> 48) against the fp64 spec written as model/pfnl.py:66-74; ragged
    tiles, one and two rows per workgroup."""
    rng = np.random.default_rng(T * 1000 + H * 10 + W)
    F = clips * T
    x = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    x64 = x.astype(np.float64)

    def check(got, ref, what):
        e = np.abs(got - ref).max()
        print(f"conv_small {what} T{T} {clips}x{H}x{W}: err {e:.3g} (|ref| <= {np.abs(ref).max():.3g})")
        assert e < 4e-6 * max(1.0, np.abs(ref).max()), (what, e)

    k1 = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b1 = (rng.normal(size=64) * 0.1).astype(np.float32)
    ref1 = pfnl_spec.lrelu(pfnl_spec.conv2d_same(x64, k1.astype(np.float64), b1.astype(np.float64)))
    got1 = ops.conv_small(dev(x), k1, b1).cpu().numpy()
    check(got1, ref1, "conv1")

    k10 = (rng.normal(size=(1, 1, 64 * T, 64)) / np.sqrt(64 * T)).astype(np.float32)
    xc = x64.reshape(clips, T, H, W, 64).transpose(0, 2, 3, 1, 4).reshape(clips, H, W, T * 64)
    ref10 = pfnl_spec.lrelu(pfnl_spec.conv2d_same(xc, k10.astype(np.float64), b1.astype(np.float64)))
    got10 = ops.conv_small(dev(x), k10, b1, b_mul=T).cpu().numpy()
    check(got10, ref10, "conv10")

    base = rng.normal(size=(clips, H, W, 64)).astype(np.float32)
    res = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    k2 = (rng.normal(size=(3, 3, 128, 64)) / 34.0).astype(np.float32)
    cat = np.concatenate([np.repeat(base.astype(np.float64), T, axis=0), x64], axis=-1)
    ref2 = pfnl_spec.lrelu(pfnl_spec.conv2d_same(cat, k2.astype(np.float64), b1.astype(np.float64))) + res
    got2 = ops.conv_small(dev(x), k2, b1, a=dev(base), a_div=T, resid=dev(res)).cpu().numpy()
    check(got2, ref2, "conv2")

    km = (rng.normal(size=(3, 3, 64 * T, 48)) / np.sqrt(9 * 64 * T)).astype(np.float32)
    bm = (rng.normal(size=48) * 0.1).astype(np.float32)
    refm = pfnl_spec.lrelu(pfnl_spec.conv2d_same(xc, km.astype(np.float64), bm.astype(np.float64)))
    gotm = ops.conv_small(dev(x), km, bm, b_mul=T).cpu().numpy()
    check(gotm[..., :48], refm, "convmerge1")
    assert not gotm[..., 48:].any()


@pytest.mark.parametrize("T,clips,H,W", [(7, 1, 32, 32), (5, 1, 64, 64), (7, 2, 10, 36), (3, 1, 7, 70), (5, 2, 33, 31), (7, 1, 1, 1)])
def test_conv_small_pf_block_two_launches(T, clips, H, W):
    """The small-shape block as the forward launches it since round 4: conv10_i has no launch - the conv1_i launch
    writes each frame's partial W10_t^T . inp1_t, the conv2_i launch sums the T partials (+ bias, leaky-relu) into its `base` source.
    Against the fp64 spec written as model/pfnl.py:66-71 (conv1 -> concat -> conv10 -> concat([base, f]) -> conv2 -> residual);
    ragged tiles, one / two / three rows per workgroup, a single pixel; repeated launches are bit-identical (fixed summation order)."""
    rng = np.random.default_rng(T * 100 + H + W)
    F = clips * T
    x = rng.normal(size=(F, H, W, 64)).astype(np.float32)
    k1 = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    k10 = (rng.normal(size=(1, 1, 64 * T, 64)) / np.sqrt(64 * T)).astype(np.float32)
    k2 = (rng.normal(size=(3, 3, 128, 64)) / 34.0).astype(np.float32)
    b1, b10, b2 = ((rng.normal(size=64) * 0.1).astype(np.float32) for _ in range(3))
    x64 = x.astype(np.float64)
    r1 = pfnl_spec.lrelu(pfnl_spec.conv2d_same(x64, k1.astype(np.float64), b1.astype(np.float64)))
    rc = r1.reshape(clips, T, H, W, 64).transpose(0, 2, 3, 1, 4).reshape(clips, H, W, T * 64)
    rb = pfnl_spec.lrelu(pfnl_spec.conv2d_same(rc, k10.astype(np.float64), b10.astype(np.float64)))
    cat = np.concatenate([np.repeat(rb, T, axis=0), r1], axis=-1)
    r2 = x64 + pfnl_spec.lrelu(pfnl_spec.conv2d_same(cat, k2.astype(np.float64), b2.astype(np.float64)))
    g1, g2 = ops.conv_small_pf_block(dev(x), k1, b1, k10, b10, k2, b2, T)
    e1, e2 = np.abs(g1.cpu().numpy() - r1).max(), np.abs(g2.cpu().numpy() - r2).max()
    print(f"conv_small block T{T} {clips}x{H}x{W}: inp1 err {e1:.3g}, out err {e2:.3g} (|ref| <= {np.abs(r2).max():.3g})")
    assert e1 < 4e-6 * max(1.0, np.abs(r1).max()) and e2 < 6e-6 * max(1.0, np.abs(r2).max())
    h1, h2 = ops.conv_small_pf_block(dev(x), k1, b1, k10, b10, k2, b2, T)
    assert torch.equal(g1, h1) and torch.equal(g2, h2)


def test_conv3x3_split16_scaling_and_data_movement():
    """Accuracy does not depend on the magnitude of the activations inside binary16's range (lo' is kept scaled by 2^11, so
    small values do not lean on binary16 subnormals); a delta kernel moves data bit-exactly (hi + lo' 2^-11 reconstructs x
    when x has <= 22 significant bits)."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 16, 32, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    base = ops.conv3x3_winograd(dev(x), k, None, act=False, variant="split16").cpu().numpy()
    ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), None)
    e0 = np.abs(base - ref).max()
    for sc in (2.0 ** -12, 2.0 ** -6, 2.0 ** 8):
        got = ops.conv3x3_winograd(dev(x * np.float32(sc)), k, None, act=False, variant="split16").cpu().numpy()
        assert np.abs(got / sc - ref).max() <= 1.5 * e0 + 1e-7, (sc, np.abs(got / sc - ref).max(), e0)
    xq = (np.round(x * 1024) / 1024).astype(np.float32)            # <= 14 significant bits
    delta = np.zeros((3, 3, 64, 64), np.float32)
    delta[0, 2, 5, 9] = 1.0                                         # out[y, x, 9] = in[y - 1, x + 1, 5]
    got = ops.conv3x3_winograd(dev(xq), delta, None, act=False, variant="split16").cpu().numpy()
    xp = np.pad(xq, ((0, 0), (1, 1), (1, 1), (0, 0)))
    assert np.array_equal(got[..., 9], xp[:, 0:16, 2:34, 5])
    assert not got[..., :9].any() and not got[..., 10:].any()


def test_split16_random_geometries():
    """Short run of tools/stress_split16.py: the f16-pipe kernels (3x3, 1x1, non-local) against the f32-MFMA path over random
    (T, clips, H, W, blocks, scale), the 3x3 op against the direct kernel, and bit-exact repeatability of every call."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_split16", os.path.join(os.path.dirname(HERE), "tools", "stress_split16.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, worst, worst_op = mod.run(seed=7, seconds=8.0)
    assert n >= 20 and worst < 2e-5 and worst_op < 2e-5


def test_round3_random_geometries():
    """Short run of tools/stress_r03.py: the split-format chain, the small-shape trunk and the default choice against the f32-MFMA
    path over random (T, clips, H, W, blocks, scale); the chain and small-shape conv2_i ops against the direct kernel; bit-exact
    repeatability of every call."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_r03", os.path.join(os.path.dirname(HERE), "tools", "stress_r03.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, worst, worst_op = mod.run(seed=11, seconds=10.0)
    assert n >= 10 and worst < 2e-5 and worst_op < 2e-5


def test_round4_kernels_random_geometries_short():
    """tools/stress_r04.py for a few seconds inside the suite (the long form ran 2 038 geometries): the second-generation bf16 3x3 kernel in
    its three modes against torch references with the same rounding points, the two-launch small-shape block against the f32-MFMA
    kernels, every call repeatable bit for bit."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("stress_r04", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_r04.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, worst = mod.run(seed=11, seconds=6.0)
    assert n >= 20 and worst < 3e-5, (n, worst)
