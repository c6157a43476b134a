import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
for (T, sc, B, H, W) in ((7, 4, 1, 32, 32), (5, 2, 1, 64, 64)):
    g = PFNLGeometry(num_frames=T, scale=sc)
    e = PFNLEngine(g); e.load_weights(synth.synthetic_weights(g, seed=0))
    x = torch.from_numpy(synth.uniform_clips(B, T, H, W, seed=1)).cuda()
    out = torch.empty(e.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    for mode in ("off", "on", "off", "on"):
        e.set_option("graph", mode)
        for _ in range(5): e.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 300
        for _ in range(n): e.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        print(T, H, W, "graph", mode, "%.4f ms" % (1e3 * (time.perf_counter() - t0) / n))
