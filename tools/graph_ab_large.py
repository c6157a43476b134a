"""hipGraph replay of the forward at the LARGE shapes (the small ones: tools/graph_ab.py): configs[1] fp32, configs[3] bf16."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
for (prec, B, H, W) in (("fp32", 4, 128, 128), ("bf16", 1, 270, 480)):
    g = PFNLGeometry()
    e = PFNLEngine(g); e.load_weights(synth.synthetic_weights(g, seed=0))
    if prec == "bf16":
        e.set_option("precision", "bf16")
    x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
    out = torch.empty(e.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    for mode in ("off", "on", "off", "on"):
        e.set_option("graph", mode)
        for _ in range(5): e.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 100
        for _ in range(n): e.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        print(prec, H, W, "graph", mode, "%.4f ms" % (1e3 * (time.perf_counter() - t0) / n))
