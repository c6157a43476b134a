"""Phase timeline of the Winograd kernel (debug build: make -C pfnl_amd/csrc dbg).
   PFNL_HIP_LIB=pfnl_amd/lib/libpfnl_hip_dbg.so python tools/wino_timing.py [fused] [ws]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops
fused = "fused" in sys.argv[1:]
variant = "winograd_ws" if "ws" in sys.argv[1:] else "winograd"
rng = np.random.default_rng(0)
F, H, W = 28, 128, 128
x = torch.from_numpy(rng.normal(size=(F, H, W, 64)).astype(np.float32)).cuda()
k = (rng.normal(size=(3, 3, 64, 64)) / 24).astype(np.float32)
kw = {}
if fused:
    kw = dict(addend=torch.zeros((4, H, W, 64), device="cuda"), add_div=7, resid=torch.zeros((F, H, W, 64), device="cuda"))
for _ in range(2):
    ops.conv3x3_winograd(x, k, None, variant=variant, **kw)
