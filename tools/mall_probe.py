"""Does the trunk's working set matter against the 256 MB Infinity Cache?  Forward time per clip / per pixel against the batch
size and the frame height, both precisions, next to the bytes a progressive-fusion block keeps live (inp0 + inp1 + base).
usage: mall_probe.py            (prints one line per shape; gpurun_out/mall_probe.txt when run through tools/run_mall_probe.sh)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth

geom = PFNLGeometry()
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))


def run(B, H, W, n):
    x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
    out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for prec, bpp in (("fp32", 256), ("bf16", 128)):
    eng.set_option("precision", prec)
    for (B, H, W) in ((1, 128, 128), (2, 128, 128), (3, 128, 128), (4, 128, 128), (6, 128, 128), (8, 128, 128), (16, 128, 128),
                      (1, 270, 480), (1, 136, 480), (2, 136, 480), (1, 68, 480), (4, 68, 480), (2, 270, 480)):
        ms = run(B, H, W, 20 if H * W * B < 100000 else 10)
        ws = (2 * 7 + 1) * B * H * W * bpp / 1e6
        print("%s B=%2d %3dx%3d: %8.3f ms/forward  %7.3f ms/clip  %8.3f ns/(clip-pixel)  live set of a block %6.0f MB"
              % (prec, B, H, W, ms, ms / B, 1e6 * ms / (B * H * W), ws), flush=True)
