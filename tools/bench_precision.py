"""fp32 vs bf16 trunk (option precision): forward time and per-class kernel time (full HIP-event profile) at
BASELINE.json configs[1] (4x7x128x128) and configs[3] (1x7x270x480 -> 1080p), device-resident input/output."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
geom = PFNLGeometry()
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
shapes = (("configs[1] 4x7x128x128", 4, 128, 128), ("configs[3] 1x7x270x480", 1, 270, 480))
for prec in sys.argv[1:] or ("fp32", "bf16"):
    eng.set_option("precision", prec)
    for name, B, H, W in shapes:
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
        eng.profile(1); eng.profile_reset()
        for _ in range(3):
            y = eng.forward(x)
        torch.cuda.synchronize()
        pr = eng.profile_read(); eng.profile(0)
        brk = {k: round(v["ms"] / 3, 3) for k, v in pr.items() if v["launches"]}
        print("%-5s %-26s %8.3f ms per forward %8.1f HR frames/s   %s" % (prec, name, ms, B / ms * 1e3, brk), flush=True)
