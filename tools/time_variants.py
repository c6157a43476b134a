"""Times the forward's kernel classes for one library variant (PFNL_HIP_LIB) - used with the -DCS_X_* timing builds."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pfnl_amd import synth  # noqa: E402
from pfnl_amd.engine import PFNLEngine  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else "split16"
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 128, 128)
geom = PFNLGeometry()
eng = PFNLEngine(geom, device=0)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
eng.set_option("conv3x3", algo)
if os.environ.get("PFNL_GRAPH"):
    eng.set_option("graph", os.environ["PFNL_GRAPH"])
xd = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=8)).cuda()
out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
for _ in range(3):
    eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
torch.cuda.synchronize()
if os.environ.get("PFNL_GRAPH"):
    t0 = time.perf_counter()
    for _ in range(20):
        eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
    torch.cuda.synchronize()
    print("graph=%s %s %dx%dx%d: %.3f ms/step" % (os.environ["PFNL_GRAPH"], algo, B, H, W, (time.perf_counter() - t0) / 20 * 1e3))
    sys.exit(0)
eng.profile_reset()
eng.profile(1)
t0 = time.perf_counter()
for _ in range(10):
    eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
eng.profile(0)
pr = eng.profile_read()
print("%-28s %s %dx%dx%d: %.3f ms/step  conv3x3 %.3f ms (%d launches, avg %.1f us)" % (
    os.path.basename(os.environ.get("PFNL_HIP_LIB", "product")), algo, B, H, W, dt * 1e3, pr["conv3x3"]["ms"] / 10,
    pr["conv3x3"]["launches"] // 10, 1e3 * pr["conv3x3"]["ms"] / max(pr["conv3x3"]["launches"], 1)))
