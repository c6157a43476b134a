"""Single-clip sharding on ONE GPU: time of each rank's strip (pfnl_forward_strip) for world sizes 1, 2, 4, 8 at 1080p -
what an N-GPU run would take per frame (max over its strips; no data-path collective), next to the full forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import dist as pd, synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
geom = PFNLGeometry()
eng = PFNLEngine(geom, device=0)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
H, W = 270, 480
x = torch.from_numpy(synth.uniform_clips(1, 7, H, W, seed=1)).cuda()
out = torch.empty(eng.out_shape(1, H, W), dtype=torch.float32, device="cuda")
for prec in ("fp32", "bf16"):
    eng.set_option("precision", prec)
    for world in (1, 2, 4, 8):
        ts = []
        for r in range(world):
            lo2, hi2 = pd.shard_range(H // 2, r, world)
            lo, hi = 2 * lo2, 2 * hi2
            for _ in range(2):
                eng.forward_strip(x, out, lo, hi - lo)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                eng.forward_strip(x, out, lo, hi - lo)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 5 * 1e3)
        print("%s 1080p, %d strips: per-strip ms %s -> frame latency %.2f ms (x%.2f)" % (prec, world, [round(t, 2) for t in ts], max(ts), ts_full / max(ts) if world > 1 else 1.0) if world > 1 or not (globals().__setitem__("ts_full", max(ts))) else "")
