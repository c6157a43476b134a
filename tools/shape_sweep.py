"""Forward time over the BASELINE.json geometries that fit fp32 (device-resident input/output, no events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
geom = PFNLGeometry()
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
for name, B, H, W in (("configs[0] 1x7x32x32", 1, 32, 32), ("configs[1] 4x7x128x128", 4, 128, 128),
                      ("16x7x128x128", 16, 128, 128), ("configs[3] geometry in fp32: 1x7x270x480 -> 1080p", 1, 270, 480),
                      ("2x7x270x480", 2, 270, 480)):
    x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
    for _ in range(3):
        y = eng.forward(x)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        y = eng.forward(x)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print("%-52s %8.3f ms per forward  %8.1f HR frames/s  workspace %.0f MB" % (name, ms, B / ms * 1e3, eng.workspace_bytes(B, H, W) / 1e6))
