"""Offline quality metrics used to produce the reference's tables — host-side
numpy, not on the hot path.

* ``psnr_rgb``      : 10*log10(1/mse) on [0,1] RGB — the in-training protocol (`model/pfnl.py:139`).
* ``rgb2ycbcr``     : ITU-R BT.601 "studio swing" (Y in 16..235), the matrix of `utils.py:194-211`
                      (= MATLAB rgb2ycbcr used by `matlab/compute_psnr.m:1-9`).
* ``psnr_y``        : `matlab/compute_psnr.m:1-18` — Y channel, boundarypixels = 0, 20*log10(255/rmse).
* ``avg_psnr``      : `utils.py:213-246` (AVG_PSNR): uint8-rounded frames, Y channel, temporal border 2,
                      spatial border 8, mean of per-frame PSNR.
* ``ssim``          : `modules/SSIM_Index.py:23-89`: 11x11 Gaussian window sigma 1.5, K = (0.01, 0.03),
                      L = 255, reflect-mode correlation (scipy.ndimage convolve default), mean SSIM map.
"""
from __future__ import annotations

import numpy as np

_T = np.array([[0.256788235294118, 0.504129411764706, 0.097905882352941],
               [-0.148223529411765, -0.290992156862745, 0.439215686274510],
               [0.439215686274510, -0.367788235294118, -0.071427450980392]])
_O = np.array([16.0, 128.0, 128.0])


def psnr_rgb(a, b) -> float:
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)


def rgb2ycbcr(img, max_val: float = 255.0) -> np.ndarray:
    """img [...,3] in [0,max_val] -> YCbCr [...,3] (Y in 16..235 when max_val = 255)."""
    x = np.asarray(img, np.float64)
    off = _O if max_val == 255.0 else _O / 255.0
    return x @ _T.T + off


def to_uint8(x, vmin=0.0, vmax=1.0) -> np.ndarray:
    x = (np.asarray(x, np.float32) - vmin) / (vmax - vmin) * 255.0
    return np.clip(np.round(x), 0, 255)


def psnr_y(img1, img2) -> float:
    """uint8-range RGB (or single-channel) images; matlab/compute_psnr.m."""
    def y(im):
        im = np.asarray(im, np.float64)
        return rgb2ycbcr(im)[..., 0] if im.ndim == 3 and im.shape[-1] == 3 else im
    d = y(img1) - y(img2)
    rmse = np.sqrt(np.mean(d ** 2))
    return float("inf") if rmse == 0 else 20.0 * np.log10(255.0 / rmse)


def avg_psnr(vid_true, vid_pred, vmin=0.0, vmax=1.0, t_border=2, sp_border=8) -> float:
    """[T,H,W,3] sequences in [vmin,vmax]; utils.py:213-246."""
    yt = np.stack([rgb2ycbcr(to_uint8(f, vmin, vmax))[..., 0] for f in vid_true])
    yp = np.stack([rgb2ycbcr(to_uint8(f, vmin, vmax))[..., 0] for f in vid_pred])
    T = yp.shape[0]
    d = (yt - yp)[t_border:T - t_border, sp_border:yp.shape[1] - sp_border, sp_border:yp.shape[2] - sp_border]
    with np.errstate(divide="ignore"):                            # identical frames: +inf, like the reference's log10(255/0)
        ps = [20.0 * np.log10(255.0 / np.sqrt(np.mean(f ** 2))) for f in d]
    return float(np.mean(ps))


def _gauss_2d(n=11, sigma=1.5) -> np.ndarray:
    m = (n - 1) / 2.0
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2.0 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h / h.sum()


def ssim(im1, im2, L: float = 255.0) -> float:
    """Single-channel images; modules/SSIM_Index.py:23-89."""
    import scipy.ndimage
    c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    w = _gauss_2d()
    a, b = np.asarray(im1, np.float64), np.asarray(im2, np.float64)
    conv = lambda z: scipy.ndimage.convolve(z, w)
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 * mu1
    s2 = conv(b * b) - mu2 * mu2
    s12 = conv(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))
    return float(np.mean(m))
