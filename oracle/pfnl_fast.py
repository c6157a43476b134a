"""FAST ORACLE / CPU BASELINE — TEST INFRASTRUCTURE ONLY.  (parity unpinned, see oracle/README.md)

torch-CPU float32 restatement of the same graph as `oracle/pfnl_spec.py` (reference:
`/root/reference/model/pfnl.py:39-80`, `/root/reference/utils.py:18-71`), using oneDNN convs and
BLAS matmuls — the stand-in for the reference's TF1.12 CPU path (Eigen/MKL-DNN), which cannot be
installed here (SURVEY.md §8(c),(d)).  It is (a) cross-checked against the fp64 spec oracle in
tests/test_oracle.py and (b) timed by bench.py's ``cpu_baseline`` leg (kind = "port").

Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def _d2s(x: torch.Tensor) -> torch.Tensor:      # NCHW, TF "DCR" order: c_in = (2i+j)*C' + k
    B, C4, H, W = x.shape
    C = C4 // 4
    x = x.view(B, 2, 2, C, H, W).permute(0, 3, 4, 1, 5, 2)   # b k h i w j
    return x.reshape(B, C, 2 * H, 2 * W)


def _s2d(x: torch.Tensor) -> torch.Tensor:      # NCHW: c_out = (dy*2+dx)*C + c
    B, C, H, W = x.shape
    x = x.view(B, C, H // 2, 2, W // 2, 2).permute(0, 3, 5, 1, 2, 4)  # b dy dx c h w
    return x.reshape(B, 4 * C, H // 2, W // 2)


def _bicubic_matrix(n_in: int, scale: int) -> torch.Tensor:
    """[n_out, n_in] float32 interpolation matrix of TF1.12's legacy bicubic (clamped taps)."""
    A = -0.75
    n_out = n_in * scale
    M = np.zeros((n_out, n_in), np.float64)
    for o in range(n_out):
        src = o / scale
        i0 = int(np.floor(src))
        t = src - i0
        d = np.array([1.0 + t, t, 1.0 - t, 2.0 - t])
        w = np.where(d <= 1.0, ((A + 2.0) * d - (A + 3.0)) * d * d + 1.0,
                     ((A * d - 5.0 * A) * d + 8.0 * A) * d - 4.0 * A)
        for k in range(4):
            M[o, min(max(i0 - 1 + k, 0), n_in - 1)] += w[k]
    return torch.from_numpy(M.astype(np.float32))


class FastOracle:
    """Weights are converted once (HWIO -> OIHW); ``forward`` is the timed CPU baseline."""

    def __init__(self, weights: Dict[str, np.ndarray], num_frames: int = 7, scale: int = 4,
                 num_block: int = 20, trunk_dtype: str = "fp32"):
        """trunk_dtype="bf16": the build-defined arithmetic of BASELINE.json configs[3] (not expressible in the
        reference, which is fp32 throughout): inside the progressive-fusion blocks every activation and kernel is
        rounded to bfloat16 (round-to-nearest-even), products are accumulated in fp32, biases stay fp32; convmerge1 reads
        the bf16 trunk with a bf16-rounded kernel and produces fp32; everything else is the fp32 graph.  Rounding points = the stores of pfnl_amd/csrc/conv_bf16.hip."""
        assert trunk_dtype in ("fp32", "bf16")
        self.T, self.scale, self.num_block = num_frames, scale, num_block
        self.trunk_bf16 = trunk_dtype == "bf16"
        self.w = {}
        for k, v in weights.items():
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
            if k.endswith("/kernel"):
                t = t.permute(3, 2, 0, 1).contiguous()
            self.w[k] = t

    def _conv(self, name, x, act=True):
        k = self.w[f"nlvsr/{name}/kernel"]
        y = F.conv2d(x, k, self.w[f"nlvsr/{name}/bias"], padding=k.shape[-1] // 2)
        return F.leaky_relu(y, 0.2) if act else y

    def _trunk_bf16(self, fr, B, T, mf, H, W):
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)              # noqa: E731  (RNE)
        fr = r(fr)
        for i in range(self.num_block):                                   # pfnl.py:65-71 with bf16 stores
            k1, b1 = r(self.w[f"nlvsr/conv1_{i}/kernel"]), self.w[f"nlvsr/conv1_{i}/bias"]
            k10, b10 = r(self.w[f"nlvsr/conv10_{i}/kernel"]), self.w[f"nlvsr/conv10_{i}/bias"]
            k2, b2 = r(self.w[f"nlvsr/conv2_{i}/kernel"]), self.w[f"nlvsr/conv2_{i}/bias"]
            a = r(F.leaky_relu(F.conv2d(fr, k1, b1, padding=1), 0.2))
            base = r(F.leaky_relu(F.conv2d(a.reshape(B, T * mf, H, W), k10, b10), 0.2))
            pb = r(F.conv2d(base, k2[:, :mf], None, padding=1))          # shared half, raw, stored in bf16
            pre = F.conv2d(a, k2[:, mf:], b2, padding=1).reshape(B, T, mf, H, W) + pb[:, None]
            fr = r(fr + F.leaky_relu(pre, 0.2).reshape(B * T, mf, H, W))
        return fr

    @torch.no_grad()
    def forward(self, x) -> np.ndarray:
        x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        B, T, H, W, c = x.shape
        assert T == self.T
        xs = x.permute(0, 1, 4, 2, 3).reshape(B, T * c, H, W)            # pfnl.py:55-56 (NCHW)
        x1 = _s2d(xs)                                                     # :57
        C = x1.shape[1]
        g = self._conv("nlblock_0/g/g", x1, act=False)                    # utils.py:26
        th = x1.reshape(B, C, -1).transpose(1, 2)                         # [B,N,C]
        f = torch.exp(th @ th.transpose(1, 2))                            # utils.py:53,57
        p = f / f.sum(-1, keepdim=True)                                   # :58
        y = (p @ g.reshape(B, C, -1).transpose(1, 2)).transpose(1, 2).reshape(B, C, H // 2, W // 2)
        z = self._conv("nlblock_0/w/w", y, act=False)                     # :67
        xs = xs + _d2s(z)                                                 # pfnl.py:59-60
        fr = xs.reshape(B * T, c, H, W)                                   # :61, frames as batch
        fr = self._conv("conv0", fr)                                      # :62
        mf = fr.shape[1]
        cen = x[:, T // 2].permute(0, 3, 1, 2)                            # :63
        My, Mx = _bicubic_matrix(H, self.scale), _bicubic_matrix(W, self.scale)
        bic = torch.einsum("oh,bchw->bcow", My, torch.einsum("pw,bchw->bchp", Mx, cen))
        if self.trunk_bf16:
            fr = self._trunk_bf16(fr, B, T, mf, H, W)
        for i in range(0 if self.trunk_bf16 else self.num_block):         # :65-71
            a = self._conv(f"conv1_{i}", fr)
            base = self._conv(f"conv10_{i}", a.reshape(B, T * mf, H, W))
            cat = torch.cat([base[:, None].expand(B, T, mf, H, W), a.reshape(B, T, mf, H, W)], 2)
            fr = fr + self._conv(f"conv2_{i}", cat.reshape(B * T, 2 * mf, H, W))
        if self.trunk_bf16:                                               # bf16 build: the merge kernel is rounded too, fp32 out
            km = self.w["nlvsr/convmerge1/kernel"].to(torch.bfloat16).to(torch.float32)
            m = F.leaky_relu(F.conv2d(fr.reshape(B, T * mf, H, W), km, self.w["nlvsr/convmerge1/bias"], padding=1), 0.2)
        else:
            m = self._conv("convmerge1", fr.reshape(B, T * mf, H, W))    # :73-74
        o = self._conv("convmerge2", _d2s(m), act=False)                  # :76-77
        if self.scale == 4:
            o = _d2s(o)                                                   # :78
        out = (o + bic).permute(0, 2, 3, 1)[:, None]                      # :80
        return out.contiguous().numpy()
