"""Single-op entry points of libpfnl_hip (the TF kernels the reference calls), on torch-ROCm
device tensors.  Used by the per-op parity tests; ``PFNLEngine.forward`` does not go through here.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi


def _req(t, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError(f"{name} must be a contiguous float32 tensor on the GPU")
    return C.c_void_p(t.data_ptr())


def _req16(t, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise TypeError(f"{name} must be a contiguous bfloat16 tensor on the GPU")
    return C.c_void_p(t.data_ptr())


def _host(a, name) -> Optional[np.ndarray]:
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _stream(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def conv2d(x, kernel, bias=None, act=True, frames_per_item=1, addend=None, add_div=1, resid=None):
    """tf.layers.Conv2D(k in {1,3}, 'same') [+ bias] [+ addend] [lrelu] [+ resid] via the MFMA kernel.

    x: [items*frames_per_item, H, W, 64] (cuda) viewed as [items, H, W, 64*fpi]; kernel: HWIO host
    array [k,k,64*fpi,cout], cout <= 64.  Reference: model/pfnl.py:49-52 applied at :66-74."""
    import torch
    lib = _capi.load_library()
    k = _host(kernel, "kernel")
    b = _host(bias, "bias")
    ks, _, cin, cout = k.shape
    F, H, W, c = x.shape
    if c != 64 or cin != 64 * frames_per_item or F % frames_per_item:
        raise ValueError("conv2d: channel / frame grouping mismatch")
    items = F // frames_per_item
    out = torch.empty((items, H, W, cout), dtype=torch.float32, device=x.device)
    _capi.check(lib.pfnl_op_conv2d(
        _req(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
        _req(addend, "addend") if addend is not None else None, int(add_div),
        _req(resid, "resid") if resid is not None else None, _req(out, "out"),
        items, frames_per_item, H, W, ks, cout, 1 if act else 0, _stream(x)))
    return out


def conv2_grouped(x, base, kernel, bias, resid, frames_per_clip, act=True):
    """conv2_i of the reference in one launch: resid + act(conv3x3(concat([base[clip], x[frame]])) + bias).
    x/resid [clips*T, H, W, 64], base [clips, H, W, 64] (cuda); kernel HWIO [3,3,128,64].
    Reference: model/pfnl.py:51, :69-71."""
    import torch
    lib = _capi.load_library()
    k = _host(kernel, "kernel")
    b = _host(bias, "bias")
    F, H, W, c = x.shape
    if k.shape != (3, 3, 128, 64) or c != 64 or F % frames_per_clip or base.shape != (F // frames_per_clip, H, W, 64):
        raise ValueError("conv2_grouped: geometry mismatch")
    out = torch.empty_like(x)
    _capi.check(lib.pfnl_op_conv2_grouped(
        _req(x, "x"), _req(base, "base"), k.ctypes.data_as(C.c_void_p),
        b.ctypes.data_as(C.c_void_p) if b is not None else None, _req(resid, "resid"), _req(out, "out"),
        F // frames_per_clip, frames_per_clip, H, W, 1 if act else 0, _stream(x)))
    return out


def conv3x3_accum(x, kernel, bias=None, act=True, frames_per_clip=1, variant="winograd"):
    """convmerge1: 3x3 over the concat of `frames_per_clip` frames, (64*fpc) -> cout <= 64, one accumulating
    launch of the persistent Winograd kernel (variant "winograd": even H, W) or of the split-f16 kernel ("split16").
    x [clips*fpc, H, W, 64] (cuda); kernel HWIO [3,3,64*fpc,cout]; returns [clips, H, W, cout].
    Reference: model/pfnl.py:52, :73-74."""
    import torch
    lib = _capi.load_library()
    k = _host(kernel, "kernel")
    b = _host(bias, "bias")
    F, H, W, c = x.shape
    cout = k.shape[3]
    if k.shape[:3] != (3, 3, 64 * frames_per_clip) or c != 64 or F % frames_per_clip or cout > 64:
        raise ValueError("conv3x3_accum: geometry mismatch")
    out = torch.empty((F // frames_per_clip, H, W, 64), dtype=torch.float32, device=x.device)
    fn = lib.pfnl_op_conv3x3_accum_split16 if variant == "split16" else lib.pfnl_op_conv3x3_accum
    _capi.check(fn(
        _req(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
        _req(out, "out"), F // frames_per_clip, frames_per_clip, H, W, cout, 1 if act else 0, _stream(x)))
    return out[..., :cout]


def conv3x3_bf16(x, kernel, bias=None, act=True, addend=None, add_div=1, resid=None):
    """bf16 trunk: 3x3 64->64 'same' convolution on bf16 MFMA, fp32 accumulation; out = act(conv + bias + addend) + resid.
    x / addend / resid / result: bfloat16 [items, H, W, 64] (cuda); kernel fp32 HWIO [3,3,64,64] (rounded inside).
    Reference: model/pfnl.py:49,51 applied at :66,69-71."""
    import torch
    lib = _capi.load_library()
    k, b = _host(kernel, "kernel"), _host(bias, "bias")
    F, H, W, c = x.shape
    if k.shape != (3, 3, 64, 64) or c != 64 or (addend is None) != (resid is None):
        raise ValueError("conv3x3_bf16: geometry mismatch")
    out = torch.empty_like(x)
    _capi.check(lib.pfnl_op_conv3x3_bf16(
        _req16(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
        _req16(addend, "addend") if addend is not None else None, add_div,
        _req16(resid, "resid") if resid is not None else None, _req16(out, "out"), F, H, W, 1 if act else 0, _stream(x)))
    return out


def conv1_conv10_bf16(x, k1, b1, k10, b10, frames_per_clip):
    """bf16 trunk: conv1_i + conv10_i in one launch.  x bfloat16 [clips*T, H, W, 64] -> (lrelu(conv3x3(x)+b1) bfloat16,
    lrelu(conv1x1 over the T outputs + b10) bfloat16 [clips, H, W, 64]).  Reference: model/pfnl.py:66-68."""
    import torch
    lib = _capi.load_library()
    k1h, b1h, k10h, b10h = _host(k1, "k1"), _host(b1, "b1"), _host(k10, "k10"), _host(b10, "b10")
    F, H, W, c = x.shape
    T = frames_per_clip
    if k1h.shape != (3, 3, 64, 64) or k10h.shape != (1, 1, 64 * T, 64) or c != 64 or F % T:
        raise ValueError("conv1_conv10_bf16: geometry mismatch")
    out1 = torch.empty_like(x)
    base = torch.empty((F // T, H, W, 64), dtype=torch.bfloat16, device=x.device)
    _capi.check(lib.pfnl_op_conv1_conv10_bf16(
        _req16(x, "x"), k1h.ctypes.data_as(C.c_void_p), b1h.ctypes.data_as(C.c_void_p) if b1h is not None else None,
        k10h.ctypes.data_as(C.c_void_p), b10h.ctypes.data_as(C.c_void_p) if b10h is not None else None,
        _req16(out1, "out1"), _req16(base, "base"), F // T, T, H, W, _stream(x)))
    return out1, base


def conv3x3_accum_bf16(x, kernel, bias=None, act=True, frames_per_clip=7):
    """bf16 trunk -> convmerge1: 3x3 over the concat of `frames_per_clip` frames, (64*fpc) -> cout <= 64, fp32 out
    [clips, H, W, 64] (channels >= cout hold act(0)).  x bfloat16 [clips*fpc, H, W, 64]; kernel fp32 HWIO [3,3,64*fpc,cout].
    Reference: model/pfnl.py:52, :73-74."""
    import torch
    lib = _capi.load_library()
    k, b = _host(kernel, "kernel"), _host(bias, "bias")
    F, H, W, c = x.shape
    T = frames_per_clip
    if k.shape[:3] != (3, 3, 64 * T) or k.shape[3] > 64 or c != 64 or F % T:
        raise ValueError("conv3x3_accum_bf16: geometry mismatch")
    out = torch.empty((F // T, H, W, 64), dtype=torch.float32, device=x.device)
    _capi.check(lib.pfnl_op_conv3x3_accum_bf16(
        _req16(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
        _req(out, "out"), F // T, T, H, W, int(k.shape[3]), 1 if act else 0, _stream(x)))
    return out


def conv1x1_bf16(x, kernel, bias=None, act=True, frames_per_item=7):
    """bf16 trunk: conv10_i, 1x1 over the concat of `frames_per_item` frames.  x bfloat16 [items*fpi, H, W, 64];
    kernel fp32 HWIO [1,1,64*fpi,64].  Reference: model/pfnl.py:50, :67-68."""
    import torch
    lib = _capi.load_library()
    k, b = _host(kernel, "kernel"), _host(bias, "bias")
    F, H, W, c = x.shape
    if k.shape != (1, 1, 64 * frames_per_item, 64) or c != 64 or F % frames_per_item:
        raise ValueError("conv1x1_bf16: geometry mismatch")
    items = F // frames_per_item
    out = torch.empty((items, H, W, 64), dtype=torch.bfloat16, device=x.device)
    _capi.check(lib.pfnl_op_conv1x1_bf16(
        _req16(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
        _req16(out, "out"), items, frames_per_item, H * W, 1 if act else 0, _stream(x)))
    return out


def conv1x1_stream(x, kernel, bias=None, act=True, frames_per_item=1, variant="stream"):
    """conv10_i: 1x1 over the concat of `frames_per_item` frames, (64*fpi) -> 64, streaming kernel.
    x: [items*fpi, H, W, 64] (cuda); kernel HWIO [1,1,64*fpi,64].  Reference: model/pfnl.py:50, :67-68."""
    import torch
    lib = _capi.load_library()
    k = _host(kernel, "kernel")
    b = _host(bias, "bias")
    F, H, W, c = x.shape
    if k.shape != (1, 1, 64 * frames_per_item, 64) or c != 64 or F % frames_per_item:
        raise ValueError("conv1x1_stream: geometry mismatch")
    items = F // frames_per_item
    out = torch.empty((items, H, W, 64), dtype=torch.float32, device=x.device)
    if variant.startswith("split16_sf"):                  # "split16_sf:io" - i, o in {0, 1}: input / output in the split format
        i_sf, o_sf = int(variant[-2]), int(variant[-1])
        _capi.check(lib.pfnl_op_conv1x1_split16_sf(
            _req(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
            _req(out, "out"), items, frames_per_item, H * W, 1 if act else 0, i_sf, o_sf, _stream(x)))
        return out
    fn = lib.pfnl_op_conv1x1_split16 if variant == "split16" else lib.pfnl_op_conv1x1_stream
    _capi.check(fn(
        _req(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
        _req(out, "out"), items, frames_per_item, H * W, 1 if act else 0, _stream(x)))
    return out


def conv_small(b, kernel, bias=None, a=None, a_div=1, b_mul=1, resid=None, items=None, act=True):
    """The small-shape trunk kernel (conv_small.hip): out[i] = act(sum_s conv(src(i, s)) + bias) (+ resid[i]) with
    src(i, s) = a[i // a_div] for s < nA (nA = 1 if `a` is given) else b[i * b_mul + s - nA]; kernel HWIO [ks, ks, 64 * nsrc, cout]."""
    import torch
    lib = _capi.load_library()
    k = _host(kernel, "kernel")
    bs = _host(bias, "bias")
    ks, _, cin, cout = k.shape
    nsrc = cin // 64
    nA = 1 if a is not None else 0
    Fb, H, W, c = b.shape
    if items is None:
        items = Fb // b_mul
    out = torch.zeros((items, H, W, 64), dtype=torch.float32, device=b.device)
    _capi.check(lib.pfnl_op_conv_small(
        _req(a, "a") if a is not None else None, _req(b, "b"), nA, int(a_div), int(b_mul), nsrc, k.ctypes.data_as(C.c_void_p),
        bs.ctypes.data_as(C.c_void_p) if bs is not None else None, _req(resid, "resid") if resid is not None else None,
        _req(out, "out"), items, H, W, ks, cout, 1 if act else 0, _stream(b)))
    return out


def conv_small_pf_block(x, k1, b1, k10, b10, k2, b2, T):
    """One progressive-fusion block on the small-shape kernels, two launches (pfnl_op_conv_small_pf_block; reference
    model/pfnl.py:66-71): x [clips*T,H,W,64] (cuda) -> (inp1, x + inp2)."""
    import torch
    lib = _capi.load_library()
    F, H, W, c = x.shape
    hs = [_host(a, n) for a, n in ((k1, "k1"), (b1, "b1"), (k10, "k10"), (b10, "b10"), (k2, "k2"), (b2, "b2"))]
    inp1 = torch.empty_like(x)
    out = torch.empty_like(x)
    _capi.check(lib.pfnl_op_conv_small_pf_block(_req(x, "x"), *[a.ctypes.data_as(C.c_void_p) for a in hs], _req(inp1, "inp1"), _req(out, "out"),
                                                F // T, int(T), H, W, _stream(x)))
    return inp1, out


def conv3x3_winograd(x, kernel, bias=None, act=True, addend=None, add_div=1, resid=None, variant="winograd"):
    """The 3x3 64->64 'same' convolution through the fused Winograd F(2x2,3x3) kernel (even H, W)."""
    import torch
    lib = _capi.load_library()
    k = _host(kernel, "kernel")
    b = _host(bias, "bias")
    if k.shape != (3, 3, 64, 64) and variant != "split16_sf_chain":
        raise ValueError("winograd path is 3x3, 64 -> 64 only")
    F, H, W, c = x.shape
    out = torch.empty((F, H, W, 64), dtype=torch.float32, device=x.device)
    if variant in ("split16_sf_in", "split16_sf_out", "split16_sf_chain"):   # the split-format kernels (conv_sf.hip / conv1_i writing SF), fp32 at this interface
        if variant != "split16_sf_chain" and k.shape != (3, 3, 64, 64):
            raise ValueError("3x3, 64 -> 64 only")
        _capi.check(lib.pfnl_op_conv3x3_split16_sf(
            {"split16_sf_in": 0, "split16_sf_out": 1, "split16_sf_chain": 2}[variant],
            _req(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
            _req(addend, "addend") if addend is not None else None, int(add_div),
            _req(resid, "resid") if resid is not None else None, _req(out, "out"), F, H, W, 1 if act else 0, _stream(x)))
        return out
    fn = {"winograd": lib.pfnl_op_conv3x3_winograd, "winograd_ws": lib.pfnl_op_conv3x3_winograd_ws,
          "split16": lib.pfnl_op_conv3x3_split16}[variant]
    _capi.check(fn(
        _req(x, "x"), k.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p) if b is not None else None,
        _req(addend, "addend") if addend is not None else None, int(add_div),
        _req(resid, "resid") if resid is not None else None, _req(out, "out"), F, H, W, 1 if act else 0, _stream(x)))
    return out


def nonlocal_residual(x, wg, bg, ww, bw, precision="fp32"):
    """x [B,T,H,W,3] (cuda) -> [B,H,W,3T] = stack + depth_to_space(NonLocalBlock(space_to_depth(stack)))
    (reference utils.py:18-71 with nltype=1, model/pfnl.py:55-60)."""
    import torch
    lib = _capi.load_library()
    B, T, H, W, c = x.shape
    C_ = 12 * T
    arrs = [_host(a, n) for a, n in ((wg, "wg"), (bg, "bg"), (ww, "ww"), (bw, "bw"))]
    if arrs[0].size != C_ * C_ or arrs[2].size != C_ * C_ or arrs[1].size != C_ or arrs[3].size != C_:
        raise ValueError("nonlocal: weight shapes do not match 12*T channels")
    out = torch.empty((B, H, W, 3 * T), dtype=torch.float32, device=x.device)
    fn = {"split16": lib.pfnl_op_nonlocal_split16, "f16": lib.pfnl_op_nonlocal_f16}.get(precision, lib.pfnl_op_nonlocal)
    _capi.check(fn(_req(x, "x"), *[a.ctypes.data_as(C.c_void_p) for a in arrs],
                                     _req(out, "out"), B, T, H, W, _stream(x)))
    return out


def bicubic(x, scale: int):
    """TF1.12 legacy ResizeBicubic (reference model/pfnl.py:63): x [B,H,W,3] (cuda) -> [B,sH,sW,3]."""
    import torch
    lib = _capi.load_library()
    B, H, W, c = x.shape
    if c != 3:
        raise ValueError("bicubic expects 3 channels")
    out = torch.empty((B, scale * H, scale * W, 3), dtype=torch.float32, device=x.device)
    _capi.check(lib.pfnl_op_bicubic(_req(x, "x"), _req(out, "out"), B, H, W, scale, _stream(x)))
    return out


def blur_decimate(hr, scale: int = 4):
    """reference utils.py:169-192 (DownSample_4D with BLUR): hr [F,H,W,3] (cuda) -> [F,ceil(H/s),ceil(W/s),3]."""
    import torch
    lib = _capi.load_library()
    F, H, W, c = hr.shape
    if c != 3:
        raise ValueError("blur_decimate expects 3 channels")
    out = torch.empty((F, -(-H // scale), -(-W // scale), 3), dtype=torch.float32, device=hr.device)
    _capi.check(lib.pfnl_op_blur_decimate(_req(hr, "hr"), _req(out, "out"), F, H, W, scale, _stream(hr)))
    return out


def selftest_mfma(device: int = 0) -> None:
    _capi.check(_capi.load_library().pfnl_selftest_mfma(device))


def _hp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def nonlocal_embedded(x, wg, bg, ww, bw, wt, bt, wp, bp):
    """The embedded-Gaussian form (reference utils.py:18-71 with nltype=0: theta = conv(x; wt, bt), phi = conv(x; wp, bp)):
    x [B,T,H,W,3] (cuda) -> [B,H,W,3T] = stack + depth_to_space(NonLocalBlock_0(space_to_depth(stack)))."""
    import torch
    lib = _capi.load_library()
    B, T, H, W, c = x.shape
    C_ = 12 * T
    arrs = [_host(a, "w") for a in (wg, bg, ww, bw, wt, bt, wp, bp)]
    for a, n in zip(arrs, (C_ * C_, C_) * 4):
        if a.size != n:
            raise ValueError("nonlocal_embedded: weight shapes do not match 12*T channels")
    out = torch.empty((B, H, W, 3 * T), dtype=torch.float32, device=x.device)
    _capi.check(lib.pfnl_op_nonlocal_embedded(_req(x, "x"), *[_hp(a) for a in arrs], _req(out, "out"), B, T, H, W, _stream(x)))
    return out


def conv1_conv10_split16(x, k1, b1, k10, b10, frames_per_clip: int, sf0: bool = False):
    """conv1_i + conv10_i of a progressive-fusion block as one launch (reference model/pfnl.py:66-68): x [clips*T,H,W,64] (cuda) ->
    (inp1 [clips*T,H,W,64], base [clips,H,W,64]), both activated; k1 HWIO [3,3,64,64], k10 HWIO [1,1,64T,64].  sf0: the input is
    converted to the split format first and the kernel takes its halo from there by LDS-DMA (bit-identical results)."""
    import torch
    lib = _capi.load_library()
    F, H, W, c = x.shape
    T = int(frames_per_clip)
    k1h, k10h = _host(k1, "k1"), _host(k10, "k10")
    if c != 64 or F % T or k1h.size != 9 * 64 * 64 or k10h.size != 64 * T * 64:
        raise ValueError("conv1_conv10_split16: geometry mismatch")
    out1 = torch.empty((F, H, W, 64), dtype=torch.float32, device=x.device)
    base = torch.empty((F // T, H, W, 64), dtype=torch.float32, device=x.device)
    fn = lib.pfnl_op_conv1_conv10_split16_sf0 if sf0 else lib.pfnl_op_conv1_conv10_split16
    _capi.check(fn(_req(x, "x"), _hp(k1h), _hp(_host(b1, "b1")), _hp(k10h), _hp(_host(b10, "b10")),
                                                 _req(out1, "out1"), _req(base, "base"), F // T, T, H, W, _stream(x)))
    return out1, base


def conv2_chain_sf0(x, kernel, bias, base, resid, add_div: int, act: bool = True):
    """The whole of conv2_i in one launch with the split-format copy of its output (reference model/pfnl.py:69-71; option split16_sf0):
    x = inp1 [F,H,W,64], kernel HWIO [3,3,128,64], base [F/add_div,H,W,64], resid [F,H,W,64] (all cuda fp32) ->
    (out [F,H,W,64] fp32, out_sf [F,H,W,128] int16 bit patterns of binary16: per pixel [channel half][hi 32 | lo' 32])."""
    import torch
    lib = _capi.load_library()
    F, H, W, c = x.shape
    k, b = _host(kernel, "kernel"), _host(bias, "bias")
    if c != 64 or k.shape != (3, 3, 128, 64) or F % add_div:
        raise ValueError("conv2_chain_sf0: geometry mismatch")
    out = torch.empty((F, H, W, 64), dtype=torch.float32, device=x.device)
    out_sf = torch.empty((F, H, W, 128), dtype=torch.int16, device=x.device)
    _capi.check(lib.pfnl_op_conv2_chain_sf0(_req(x, "x"), _hp(k), _hp(b) if b is not None else None, _req(base, "base"), int(add_div),
                                            _req(resid, "resid"), _req(out, "out"), C.c_void_p(out_sf.data_ptr()), F, H, W, 1 if act else 0, _stream(x)))
    return out, out_sf


def nonlocal_block(x, wg, bg, ww, bw, theta=None, phi=None, nltype: int = 1, sub_sample: int = 1):
    """utils.NonLocalBlock(input_x, out_channels, sub_sample, nltype) in its general form (reference utils.py:18-71) inside the
    wrapper of model/pfnl.py:55-60: x [B,T,H,W,3] (cuda) -> [B,H,W,3T] = stack + depth_to_space(NonLocalBlock(space_to_depth(stack))).
    ``theta`` / ``phi`` = (kernel [C,C] or [1,1,C,C], bias [C]) of the two projections (nltype 0 and 2)."""
    import torch
    lib = _capi.load_library()
    B, T, H, W, c = x.shape
    C_ = 12 * T
    if nltype not in (0, 1, 2):
        raise ValueError("nltype must be 0, 1 or 2 (3, 'concat', builds no graph in the reference: utils.py:23)")
    if nltype != 1 and (theta is None or phi is None):
        raise ValueError("nltype 0 / 2 need the theta and phi projections")
    arrs = [_host(a, "w") for a in (wg, bg, ww, bw)]
    proj = [None] * 4
    if nltype != 1:
        proj = [_host(a, "w") for a in (theta[0], theta[1], phi[0], phi[1])]
    for a, n in zip(arrs + [p for p in proj if p is not None], (C_ * C_, C_) * 4):
        if a.size != n:
            raise ValueError("nonlocal_block: weight shapes do not match 12*T channels")
    out = torch.empty((B, H, W, 3 * T), dtype=torch.float32, device=x.device)
    _capi.check(lib.pfnl_op_nonlocal_block(_req(x, "x"), *[_hp(a) for a in arrs], *[_hp(a) for a in proj], int(nltype),
                                           int(sub_sample), _req(out, "out"), B, T, H, W, _stream(x)))
    return out


def conv0(x, kernel, bias=None):
    """conv0 (reference model/pfnl.py:48,61-62): x [B,T,H,W,3] (cuda) -> lrelu(conv5x5(frame) + b) [B*T,H,W,64]."""
    import torch
    lib = _capi.load_library()
    B, T, H, W, c = x.shape
    k, b = _host(kernel, "kernel"), _host(bias, "bias")
    if k.shape != (5, 5, 3, 64) or c != 3:
        raise ValueError("conv0: geometry mismatch")
    out = torch.empty((B * T, H, W, 64), dtype=torch.float32, device=x.device)
    _capi.check(lib.pfnl_op_conv0(_req(x, "x"), _hp(k), _hp(b), _req(out, "out"), B, T, H, W, _stream(x)))
    return out


def tail(merge, x, kernel, bias, scale: int):
    """The tail (reference model/pfnl.py:53,63,76-80): merge [B,H,W,48] (cuda), x [B,T,H,W,3] (cuda) -> [B,1,sH,sW,3]."""
    import torch
    lib = _capi.load_library()
    B, T, H, W, c = x.shape
    k, b = _host(kernel, "kernel"), _host(bias, "bias")
    if merge.shape != (B, H, W, 48) or k.shape != (3, 3, 12, 12 if scale == 4 else 3):
        raise ValueError("tail: geometry mismatch")
    out = torch.empty((B, 1, scale * H, scale * W, 3), dtype=torch.float32, device=x.device)
    _capi.check(lib.pfnl_op_tail(_req(merge, "merge"), _req(x, "x"), _hp(k), _hp(b), _req(out, "out"), B, T, H, W, scale, _stream(x)))
    return out


def gather_windows(frames, first: int, count: int, num_frames: int):
    """frames [F,H,W,3] (cuda) -> [count,T,H,W,3]: the clamped sliding windows of frames first..first+count-1
    (reference model/pfnl.py:238-242)."""
    import torch
    lib = _capi.load_library()
    F, H, W, c = frames.shape
    if c != 3:
        raise ValueError("gather_windows expects [F,H,W,3]")
    out = torch.empty((count, num_frames, H, W, 3), dtype=torch.float32, device=frames.device)
    _capi.check(lib.pfnl_op_gather_windows(_req(frames, "frames"), _req(out, "out"), F, first, count, num_frames, H, W, _stream(frames)))
    return out


def quantise_u8(sr):
    """uint8(np.round(np.clip(sr * 255, 0, 255))) on the device (reference model/pfnl.py:254-257)."""
    import torch
    lib = _capi.load_library()
    _req(sr, "sr")
    out = torch.empty(sr.shape, dtype=torch.uint8, device=sr.device)
    _capi.check(lib.pfnl_op_quantise_u8(C.c_void_p(sr.data_ptr()), C.c_void_p(out.data_ptr()), sr.numel(), _stream(sr)))
    return out
