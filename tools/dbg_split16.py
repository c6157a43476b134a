import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pfnl_spec
from pfnl_amd import ops
rng = np.random.default_rng(0)
items, H, W = 1, 8, 32
x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
b = np.zeros(64, np.float32)
ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
xd = torch.from_numpy(x).cuda()
for mode in ("plain", "fused"):
    kw = {}
    if mode == "fused":
        kw = dict(addend=torch.zeros_like(xd), add_div=1, resid=torch.zeros_like(xd))
    got = ops.conv3x3_winograd(xd, k, b, act=False, variant="split16", **kw).cpu().numpy()
    e = np.abs(got - ref)
    print(mode, "max err", e.max(), " rows", np.round(e.max(axis=(0, 2, 3)), 2), " ch/4", np.round(e.reshape(1, H, W, 16, 4).max(axis=(0, 1, 2, 4)), 2))
    print("  got[0,2,5,44:64] ", np.round(got[0, 2, 5, 44:64], 3))
    print("  ref[0,2,5,44:64] ", np.round(ref[0, 2, 5, 44:64], 3))
    # is the wrong data some other pixel's / channel's correct value?
    bad = got[0, 2, 5, 48:52]
    d = np.abs(ref[0][..., None] - bad[None, None, None, :]).reshape(-1, 4)
    idx = np.argmin(d, axis=0)
    print("  nearest ref entries for got[0,2,5,48:52]:", [np.unravel_index(i, ref[0].shape) for i in idx], np.round(d.min(axis=0), 4))
