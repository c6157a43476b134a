"""Rare-event check (GPU): the 3x3 kernels are deterministic, so N repetitions of one launch must be BIT-identical, and the
first one must agree with the direct f32-MFMA kernel.  Written after a store-data hazard (pfnl_amd/csrc/common.h,
buffer_store_b128_guarded) showed up as one corrupted dword in 16 lanes once in ~10^5 stores.

usage: python tools/stress_repeat.py [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pfnl_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(5)
bad = 0
for items, H, W in ((21, 16, 32), (7, 33, 70), (28, 128, 128), (7, 270, 480)):
    x = torch.from_numpy(rng.normal(size=(items, H, W, 64)).astype(np.float32)).cuda()
    add = torch.from_numpy(rng.normal(size=(items // 7, H, W, 64)).astype(np.float32)).cuda()
    res = torch.from_numpy(rng.normal(size=(items, H, W, 64)).astype(np.float32)).cuda()
    k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    for fused in (True, False):
        kw = dict(addend=add, add_div=7, resid=res) if fused else {}
        ref = ops.conv2d(x, k, b, act=True, **kw)
        for variant in ("split16", "winograd_ws") if H % 2 == 0 and W % 2 == 0 else ("split16",):
            first = ops.conv3x3_winograd(x, k, b, act=True, variant=variant, **kw)
            err = float((first - ref).abs().max())
            diff = 0
            n = reps if H < 200 else max(reps // 10, 5)
            for _ in range(n):
                again = ops.conv3x3_winograd(x, k, b, act=True, variant=variant, **kw)
                diff += int((again != first).sum())
            ok = diff == 0 and err < 2e-5
            bad += not ok
            print("%-12s %2dx%3dx%3d fused %d: %4d repetitions, %d differing elements, max|.-direct| %.2e %s" % (
                variant, items, H, W, fused, n, diff, err, "" if ok else "  <-- FAIL"))
print("stress_repeat:", "ok" if not bad else "%d FAILURES" % bad)
sys.exit(1 if bad else 0)
