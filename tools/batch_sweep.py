"""Forward time per clip against the batch size (configs[1] geometry): does the trunk's working set (234 MB per 4 clips of
inp0 + inp1) matter against the 256 MB Infinity Cache?  usage: batch_sweep.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
geom = PFNLGeometry()
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
for B in (1, 2, 3, 4, 6, 8, 12, 16):
    x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
    out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    for _ in range(3):
        eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
    torch.cuda.synchronize()
    n = 12
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print("B=%2d %dx%d: %7.3f ms per forward, %6.3f ms per clip, %7.1f HR frames/s, trunk pair %.0f MB" % (B, H, W, ms, ms / B, B / ms * 1e3, 2 * B * 7 * H * W * 256 / 1e6))
