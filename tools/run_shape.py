"""Run a few forwards of one shape / precision (for rocprofv3): python tools/run_shape.py bf16 1 270 480 [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
prec, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
geom = PFNLGeometry()
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
eng.set_option("precision", prec)
x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
for _ in range(iters):
    y = eng.forward(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    y = eng.forward(x)
torch.cuda.synchronize()
print("%s %dx7x%dx%d: %.3f ms per forward" % (prec, B, H, W, 1e3 * (time.perf_counter() - t0) / iters))
