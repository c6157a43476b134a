"""configs[0] / configs[4] forward time without profiling events: eager against hipGraph replay.  usage: small_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
for name, g, B, H, W in (("configs[0] 1x7x32x32", PFNLGeometry(), 1, 32, 32), ("configs[4] 1x5x64x64 2x", PFNLGeometry(num_frames=5, scale=2, num_block=20), 1, 64, 64),
                         ("1x7x64x64", PFNLGeometry(), 1, 64, 64)):
    for graph in ("off", "on"):
        for small in ("auto", "off"):
            eng = PFNLEngine(g)
            eng.load_weights(synth.synthetic_weights(g, seed=0))
            eng.set_option("graph", graph)
            eng.set_option("small", small)
            x = torch.from_numpy(synth.uniform_clips(B, g.num_frames, H, W, seed=1)).cuda()
            out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(5):
                eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, st)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n):
                eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, st)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            print("%-26s graph=%-3s small=%-4s %7.4f ms per forward  %8.1f HR frames/s" % (name, graph, small, ms, B / ms * 1e3))
            eng.close()
