"""SPEC ORACLE — TEST INFRASTRUCTURE ONLY.  (parity unpinned, see oracle/README.md)

A naive numpy float64 restatement of the reference's PFNL forward graph, written op-for-op from
`/root/reference/model/pfnl.py:39-80` (``PFNL.forward``) and `/root/reference/utils.py:18-71`
(``NonLocalBlock``: all the branches that build a graph - nltype 0 / 1 / 2, sub_sample), with the TensorFlow 1.12 op semantics the reference relies on
restated from their published definitions (the arithmetic lives in TF 1.12.0, named at the
reference's `README.md:23`, not vendored and not installable here):

* ``Conv2D(padding='same', strides=1)``: cross-correlation, HWIO kernels, symmetric k//2 zero pad.
* ``tf.nn.leaky_relu``: max(0.2*x, x)                       (`model/pfnl.py:42`)
* ``tf.space_to_depth(x,2)``: out[b,h,w,(dy*2+dx)*C+c] = x[b,2h+dy,2w+dx,c]   (`pfnl.py:57`)
* ``tf.depth_to_space(x,2)``: out[b,2h+i,2w+j,k] = x[b,h,w,(2i+j)*C'+k]       (`pfnl.py:59,76,78`)
* ``tf.image.resize_images(..., method=2)``: TF1 legacy bicubic, align_corners=False, no half-pixel
  centres, Keys A=-0.75, taps clamped to the border, no renormalisation           (`pfnl.py:63`)
* affinity as written: exp, sum, divide — NO max subtraction            (`utils.py:57-58`)
* ``tf.layers.average_pooling2d(pool, strides=pool)``: padding 'valid' (its default)  (`utils.py:27-28,35-36`)

Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

LRELU_ALPHA = 0.2  # tf.nn.leaky_relu default, model/pfnl.py:42


def lrelu(x: np.ndarray) -> np.ndarray:
    return np.maximum(LRELU_ALPHA * x, x)


def conv2d_same(x: np.ndarray, kernel: np.ndarray, bias: Optional[np.ndarray]) -> np.ndarray:
    """x [B,H,W,Cin], kernel [kh,kw,Cin,Cout] (HWIO), stride 1, SAME (odd k)."""
    kh, kw, cin, cout = kernel.shape
    assert kh % 2 == 1 and kw % 2 == 1 and x.shape[-1] == cin
    ph, pw = kh // 2, kw // 2
    xp = np.pad(x, ((0, 0), (ph, ph), (pw, pw), (0, 0)))
    B, H, W, _ = x.shape
    out = np.zeros((B, H, W, cout), dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            out += xp[:, dy:dy + H, dx:dx + W, :] @ kernel[dy, dx]
    if bias is not None:
        out = out + bias
    return out


def space_to_depth2(x: np.ndarray) -> np.ndarray:
    B, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0, "space_to_depth(2) needs even H and W (model/pfnl.py:57)"
    x = x.reshape(B, H // 2, 2, W // 2, 2, C)          # b h dy w dx c
    x = x.transpose(0, 1, 3, 2, 4, 5)                   # b h w dy dx c
    return x.reshape(B, H // 2, W // 2, 4 * C)


def depth_to_space2(x: np.ndarray) -> np.ndarray:
    B, H, W, C4 = x.shape
    assert C4 % 4 == 0
    C = C4 // 4
    x = x.reshape(B, H, W, 2, 2, C)                     # b h w i j k
    x = x.transpose(0, 1, 3, 2, 4, 5)                   # b h i w j k
    return x.reshape(B, 2 * H, 2 * W, C)


def _bicubic_taps(t: float, A: float = -0.75):
    # Keys cubic convolution kernel, distances 1+t, t, 1-t, 2-t
    def near(d):
        return ((A + 2.0) * d - (A + 3.0)) * d * d + 1.0

    def far(d):
        return ((A * d - 5.0 * A) * d + 8.0 * A) * d - 4.0 * A

    return np.array([far(1.0 + t), near(t), near(1.0 - t), far(2.0 - t)])


def resize_bicubic_tf1(x: np.ndarray, scale: int) -> np.ndarray:
    """TF1.12 ResizeBicubic (align_corners=False), integer upscale.  x [B,H,W,C]."""
    B, H, W, C = x.shape

    def axis_tables(n_in, n_out):
        idx = np.empty((n_out, 4), np.int64)
        wts = np.empty((n_out, 4))
        for o in range(n_out):
            src = o * (n_in / n_out)
            i0 = int(np.floor(src))
            t = src - i0
            idx[o] = np.clip(np.arange(i0 - 1, i0 + 3), 0, n_in - 1)
            wts[o] = _bicubic_taps(t)
        return idx, wts

    iy, wy = axis_tables(H, H * scale)
    ix, wx = axis_tables(W, W * scale)
    # x direction first, then y (the order of TF's kernel; irrelevant in exact arithmetic)
    tmp = np.zeros((B, H, W * scale, C), dtype=x.dtype)
    for k in range(4):
        tmp += x[:, :, ix[:, k], :] * wx[None, None, :, k, None]
    out = np.zeros((B, H * scale, W * scale, C), dtype=x.dtype)
    for k in range(4):
        out += tmp[:, iy[:, k], :, :] * wy[None, :, k, None, None]
    return out


def avg_pool_valid(x: np.ndarray, k: int) -> np.ndarray:
    """tf.layers.average_pooling2d(pool_size=k, strides=k) with its default padding 'valid' (utils.py:27-28, 35-36):
    x [B,h,w,C] -> [B,h//k,w//k,C]; rows / columns that do not fill a window are dropped."""
    B, h, w, C = x.shape
    hp, wp = h // k, w // k
    return x[:, :hp * k, :wp * k].reshape(B, hp, k, wp, k, C).mean(axis=(2, 4))


def nonlocal_block(x: np.ndarray, wg, bg, ww, bw, stabilise: bool = False, theta=None, phi=None,
                   nltype: Optional[int] = None, sub_sample: int = 1) -> np.ndarray:
    """utils.py:18-71.  x [B,h,w,C] -> [B,h,w,C] (no residual, :70).
    nltype=1 (PFNL's call, model/pfnl.py:58): theta = phi = x.  nltype=0 (embedded Gaussian; the option north_star
    names) and nltype=2 (dot product): ``theta`` / ``phi`` = (kernel [1,1,C,C], bias [C]) of the two extra 1x1 convs
    (:31-32, :39-40).  ``nltype`` None: 0 when theta / phi are given, else 1.  sub_sample > 1: g and phi average-pooled
    (:27-28, :35-36).  nltype 3 ('concat') builds no graph in the reference (the function falls off its ``if nltype<=2``)."""
    if nltype is None:
        nltype = 1 if theta is None else 0
    if nltype not in (0, 1, 2):
        raise ValueError("nltype 3 builds no graph in the reference (utils.py:23)")
    B, h, w, C = x.shape
    g = conv2d_same(x, wg, bg)                           # utils.py:26
    if sub_sample > 1:
        g = avg_pool_valid(g, sub_sample)                # :27-28
    ph = x if nltype == 1 else conv2d_same(x, phi[0], phi[1])            # :31-34
    if sub_sample > 1:
        ph = avg_pool_valid(ph, sub_sample)              # :35-36
    th = x if nltype == 1 else conv2d_same(x, theta[0], theta[1])        # :39-42
    g_x = g.reshape(B, -1, C)                            # :44
    theta_x = th.reshape(B, -1, C)                       # :45
    phi_x = ph.reshape(B, -1, C).transpose(0, 2, 1)      # :49-50
    f = theta_x @ phi_x                                  # :53
    if nltype <= 1:
        if stabilise:
            f = f - f.max(axis=-1, keepdims=True)
        f = np.exp(f)                                    # :57
        f_softmax = f / f.sum(axis=-1, keepdims=True)    # :58
    else:
        f = np.maximum(f, 0)                             # :60
        with np.errstate(invalid="ignore", divide="ignore"):
            f_softmax = f / f.sum(axis=2, keepdims=True)  # :61-63 (0 / 0 = NaN for a query without a positive affinity, as in TF)
    y = f_softmax @ g_x                                  # :64
    y = y.reshape(B, h, w, C)                            # :65
    return conv2d_same(y, ww, bw)                        # :67


def forward(x: np.ndarray, weights: Dict[str, np.ndarray], scale: int = 4, num_block: int = 20,
            dtype=np.float64, stabilise: bool = False, taps: Optional[dict] = None,
            nltype: Optional[int] = None, sub_sample: int = 1) -> np.ndarray:
    """model/pfnl.py:39-80.  x [B,T,H,W,3] in [0,1] -> [B,1,scale*H,scale*W,3].

    ``taps``: optional dict that receives named intermediates (for per-op parity tests)."""
    x = np.asarray(x, dtype=dtype)
    Wt = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
    B, T, H, W, c = x.shape                              # pfnl.py:44 (its "w,h" are really H,W)

    def conv(name, inp, act=True):
        y = conv2d_same(inp, Wt[f"nlvsr/{name}/kernel"], Wt[f"nlvsr/{name}/bias"])
        return lrelu(y) if act else y

    inp0 = np.concatenate([x[:, i] for i in range(T)], axis=-1)          # :55-56
    inp1 = space_to_depth2(inp0)                                          # :57
    tp = [None, None]                                                     # nltype 0 option: theta / phi variables present
    for j, n in enumerate(("theta/theta", "phi/phi")):
        if f"nlvsr/nlblock_0/{n}/kernel" in Wt:
            tp[j] = (Wt[f"nlvsr/nlblock_0/{n}/kernel"], Wt[f"nlvsr/nlblock_0/{n}/bias"])
    inp1 = nonlocal_block(inp1, Wt["nlvsr/nlblock_0/g/g/kernel"], Wt["nlvsr/nlblock_0/g/g/bias"],
                          Wt["nlvsr/nlblock_0/w/w/kernel"], Wt["nlvsr/nlblock_0/w/w/bias"],
                          stabilise=stabilise, theta=tp[0], phi=tp[1],    # :58 (the reference pins nltype=1, sub_sample=1;
                          nltype=nltype, sub_sample=sub_sample)           #      the two arguments are utils.NonLocalBlock's)
    inp1 = depth_to_space2(inp1)                                          # :59
    inp0 = inp0 + inp1                                                    # :60
    if taps is not None:
        taps["nl_out"] = inp0.copy()
    frames = np.split(inp0, T, axis=-1)                                   # :61
    frames = [conv("conv0", f) for f in frames]                           # :62
    if taps is not None:
        taps["conv0"] = np.stack(frames, 1)
    bic = resize_bicubic_tf1(x[:, T // 2], scale)                         # :63
    if taps is not None:
        taps["bic"] = bic.copy()

    for i in range(num_block):                                            # :65
        inp1 = [conv(f"conv1_{i}", f) for f in frames]                    # :66
        base = np.concatenate(inp1, axis=-1)                              # :67
        base = conv(f"conv10_{i}", base)                                  # :68
        inp2 = [np.concatenate([base, f], -1) for f in inp1]              # :69
        inp2 = [conv(f"conv2_{i}", f) for f in inp2]                      # :70
        frames = [frames[j] + inp2[j] for j in range(T)]                  # :71
        if taps is not None and i == 0:
            taps["block0"] = np.stack(frames, 1)

    if taps is not None:
        taps["trunk"] = np.stack(frames, 1)
    merge = np.concatenate(frames, axis=-1)                               # :73
    merge = conv("convmerge1", merge)                                     # :74
    if taps is not None:
        taps["merge1"] = merge.copy()
    large1 = depth_to_space2(merge)                                       # :76
    out1 = conv("convmerge2", large1, act=False)                          # :77
    out = depth_to_space2(out1) if scale == 4 else out1                   # :78 (2x: build-defined)
    return np.stack([out + bic], axis=1)                                  # :80


# ---- harness pieces (model/pfnl.py:236-258) -------------------------------------------------

def sliding_windows(lrs: np.ndarray, num_frames: int = 7) -> np.ndarray:
    """[F,H,W,3] -> [F,T,H,W,3] with border-clamped indices (`model/pfnl.py:238-242`)."""
    F = lrs.shape[0]
    half = num_frames // 2
    out = []
    for i in range(F):
        index = np.clip(np.arange(i - half, i + half + 1), 0, F - 1)
        out.append(lrs[index])
    return np.stack(out, 0) if out else np.zeros((0, num_frames) + lrs.shape[1:], lrs.dtype)


def quantise(sr: np.ndarray) -> np.ndarray:
    """`model/pfnl.py:254-257`: *255, clip, np.round (half-to-even), uint8."""
    img = np.asarray(sr) * 255.0
    img = np.clip(img, 0, 255)
    return np.round(img, 0).astype(np.uint8)
