"""The two MFMA shapes of the chain launch of conv2_i (fp32 path; option split16_mfma = 16 | 32) at a few shapes, alternating, ms per forward
(device-resident input and output).  usage: python tools/split16_mfma_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
g = PFNLGeometry()
e = PFNLEngine(g)
e.load_weights(synth.synthetic_weights(g, seed=0))
pass
for (B, H, W) in ((4, 128, 128), (1, 270, 480), (1, 180, 318), (8, 128, 128), (5, 128, 128)):
    x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
    out = torch.empty(e.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    res = []
    for m in ("16", "32", "16", "32"):
        e.set_option("split16_mfma", m)
        for _ in range(10):
            e.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 150
        for _ in range(n):
            e.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        res.append("%s: %.3f" % (m, 1e3 * (time.perf_counter() - t0) / n))
    print("%dx7x%dx%d fp32 (%s): ms per forward with split16_mfma = %s" % (B, H, W, e.plan(B, H, W)["structure"], ", ".join(res)), flush=True)
