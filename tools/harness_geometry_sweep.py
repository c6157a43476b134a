"""HR frames/s at the geometries the reference's own harness runs (Vid4 LR 144x180 / 144x176 / 120x180, UDM10 LR
180x318, the eval batch 4x128x240, `part`-sized batches of windows) next to the BASELINE.json configs, fp32 and bf16 trunk, input and result
resident in HBM.  The batch of a sequence = num_once windows (model/pfnl.py:236-247 with part=50: calendar 41 frames -> 41 in one batch,
city 34, foliage 49, walk 47; UDM10 32 each).  usage: python tools/harness_geometry_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth

CASES = (("configs[0] 1x7x32x32", 1, 32, 32), ("configs[1] 4x7x128x128", 4, 128, 128), ("eval batch 4x7x128x240", 4, 128, 240),
         ("Vid4 calendar 41x7x144x180 (one `part`)", 41, 144, 180), ("Vid4 city 34x7x144x176", 34, 144, 176),
         ("Vid4 foliage 49x7x120x180", 49, 120, 180), ("Vid4 one window 1x7x144x180", 1, 144, 180), ("UDM10 32x7x180x318", 32, 180, 318), ("UDM10 one window 1x7x180x318", 1, 180, 318),
         ("configs[3] 1x7x270x480 -> 1080p", 1, 270, 480))

geom = PFNLGeometry()
for prec in ("fp32", "bf16"):
    eng = PFNLEngine(geom)
    eng.load_weights(synth.synthetic_weights(geom, seed=0))
    eng.set_option("precision", prec)
    for name, B, H, W in CASES:
        x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
        out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
        for _ in range(2):
            eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        n = 20 if B * H * W < 300000 else 5
        t0 = time.perf_counter()
        for _ in range(n):
            eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print("%s %-44s %9.3f ms per forward %8.1f HR frames/s  %6.3f us per LR pixel and clip  workspace %6.0f MB"
              % (prec, name, ms, B / ms * 1e3, ms * 1e3 / (B * H * W), eng.workspace_bytes(B, H, W) / 1e6), flush=True)
        if B > 4:   # a `part`-sized batch spans > 2^31 bytes per activation: its first / last clips against single-clip forwards
            ref = torch.empty(eng.out_shape(1, H, W), dtype=torch.float32, device="cuda")
            for k in (0, B // 2, B - 1):
                eng.forward_device(x[k:k + 1].contiguous().data_ptr(), ref.data_ptr(), 1, H, W, 0)
                torch.cuda.synchronize()
                d = float((out[k:k + 1] - ref).abs().max())
                assert d < (2e-5 if prec == "fp32" else 2e-2), (name, k, d)
            print("     clips 0, %d, %d of the batch = their single-clip forwards (max |d| %.1e)" % (B // 2, B - 1, d), flush=True)
            del ref
        del x, out
    eng.close()
