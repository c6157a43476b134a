"""Seeded synthetic weights and clips: there is no network, so neither the
pretrained checkpoint (`checkpoint/README.md` in the reference) nor Vid4/UDM10 can be fetched.

* weights: Xavier-uniform kernels like the reference's initialiser (`model/pfnl.py:45`; the two
  non-local 1x1 convs use TF's default glorot-uniform, `utils.py:26,67`), with ``conv2_i`` scaled by
  0.1 and ``convmerge2`` by 0.25 so that the 20-block residual trunk stays O(1) and the network
  output is "bicubic + small residual" like a trained model; biases U(-0.01, 0.01) so that every
  bias path is exercised (the reference initialises them to zero).
* clips: (a) U[0,1) noise; (b) a band-limited moving random field blurred+decimated with the
  reference's 13x13 sigma=1.6 Gaussian (`utils.py:95-105,169-192`), which also yields a ground
  truth HR centre frame for the PSNR-parity metric.

numpy only; deterministic for a given seed on every machine.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .spec import PFNLGeometry


def synthetic_weights(geom: PFNLGeometry = PFNLGeometry(), seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in geom.weight_shapes():
        if name.endswith("/kernel"):
            kh, kw, cin, cout = shape
            fan_in, fan_out = kh * kw * cin, kh * kw * cout
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-lim, lim, size=shape)
            layer = name.split("/")[1]
            if layer.startswith("conv2_"):
                w *= 0.1
            elif layer == "convmerge2":
                w *= 0.25
            out[name] = w.astype(np.float32)
        else:
            out[name] = rng.uniform(-0.01, 0.01, size=shape).astype(np.float32)
    return out


def uniform_clips(B: int, T: int, H: int, W: int, seed: int = 1234) -> np.ndarray:
    """[B,T,H,W,3] float32 U[0,1)."""
    rng = np.random.default_rng(seed)
    return rng.random((B, T, H, W, 3), dtype=np.float32)


def gaussian_kernel_1d(kernlen: int = 13, nsig: float = 1.6) -> np.ndarray:
    """1-D factor of the reference's BLUR (`utils.py:95-105`): scipy's gaussian_filter of a delta is
    the outer product of this normalised, truncated Gaussian."""
    r = kernlen // 2
    i = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-0.5 * (i / nsig) ** 2)
    return k / k.sum()


def blur_decimate(hr: np.ndarray, scale: int = 4) -> np.ndarray:
    """[..., H, W, C] -> [..., H/scale, W/scale, C]: reflect-pad 6/6 (edge not repeated), 13x13
    Gaussian, stride ``scale``, VALID (`utils.py:169-192`)."""
    k = gaussian_kernel_1d()
    r = len(k) // 2
    x = np.asarray(hr, dtype=np.float64)
    lead = x.shape[:-3]
    x = x.reshape((-1,) + x.shape[-3:])
    xp = np.pad(x, ((0, 0), (r, r), (r, r), (0, 0)), mode="reflect")
    H, W = x.shape[1], x.shape[2]
    oh = (H + 2 * r - len(k)) // scale + 1
    ow = (W + 2 * r - len(k)) // scale + 1
    tmp = np.zeros((x.shape[0], oh, xp.shape[2], x.shape[3]))
    for i, kv in enumerate(k):
        tmp += kv * xp[:, i:i + (oh - 1) * scale + 1:scale]
    out = np.zeros((x.shape[0], oh, ow, x.shape[3]))
    for i, kv in enumerate(k):
        out += kv * tmp[:, :, i:i + (ow - 1) * scale + 1:scale]
    return out.reshape(lead + out.shape[1:]).astype(np.float32)


def moving_field_clips(B: int, T: int, H: int, W: int, scale: int = 4, seed: int = 1234,
                       n_waves: int = 24) -> Tuple[np.ndarray, np.ndarray]:
    """Band-limited random field with per-frame sub-pixel translation (<= 2 HR px / frame).

    Returns (lr [B,T,H,W,3] float32, gt [B,scale*H,scale*W,3] float32 = centre HR frame)."""
    rng = np.random.default_rng(seed)
    HH, WW = H * scale, W * scale
    yy, xx = np.meshgrid(np.arange(HH, dtype=np.float64), np.arange(WW, dtype=np.float64), indexing="ij")
    lr = np.empty((B, T, H, W, 3), np.float32)
    gt = np.empty((B, HH, WW, 3), np.float32)
    for b in range(B):
        fx = rng.uniform(-0.35, 0.35, size=(n_waves, 3))
        fy = rng.uniform(-0.35, 0.35, size=(n_waves, 3))
        ph = rng.uniform(0, 2 * np.pi, size=(n_waves, 3))
        amp = rng.uniform(0.2, 1.0, size=(n_waves, 3)) / np.sqrt(n_waves)
        vel = rng.uniform(-2.0, 2.0, size=2)
        frames = np.empty((T, HH, WW, 3))
        for t in range(T):
            dy, dx = vel[0] * (t - T // 2), vel[1] * (t - T // 2)
            for c in range(3):
                acc = np.zeros((HH, WW))
                for k in range(n_waves):
                    acc += amp[k, c] * np.sin(fy[k, c] * (yy + dy) + fx[k, c] * (xx + dx) + ph[k, c])
                frames[t, :, :, c] = acc
        frames = np.clip(0.5 + 0.35 * frames, 0.0, 1.0)
        lr[b] = blur_decimate(frames, scale)
        gt[b] = frames[T // 2].astype(np.float32)
    return lr, gt


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    """10*log10(1/mse) on [0,1] RGB, as `model/pfnl.py:139`."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    if mse == 0.0:
        return float("inf")
    return 10.0 * np.log10(1.0 / mse)
