import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops
from oracle import pfnl_spec
rng = np.random.default_rng(0)
for (items, H, W) in [(1, 4, 32), (2, 4, 32), (1, 8, 64)]:
    x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
    for name in ("center", "random"):
        k = np.zeros((3, 3, 64, 64), np.float32)
        if name == "center":
            for c in range(64): k[1, 1, c, c] = 1.0
        else:
            k = (rng.normal(size=(3, 3, 64, 64)) / 24).astype(np.float32)
        ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), None)
        got = ops.conv3x3_winograd(torch.from_numpy(x).cuda(), k, None, act=False, variant="winograd16").cpu().numpy()
        err = np.abs(got - ref)
        print(items, H, W, name, "max", err.max(), "| by item", err.max(axis=(1,2,3)), "| by y%2", [err[:, i::2].max() for i in range(2)],
              "| by x%2", [err[:, :, i::2].max() for i in range(2)], "| by cout half", err[..., :32].max(), err[..., 32:].max(),
              "| by y", err.max(axis=(0,2,3)).round(3))
