"""How long does the host need to ENQUEUE one forward (all kernel launches) compared with its GPU time?
(If enqueue time approached GPU time, 8 ranks sharing one host would become launch-bound.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
geom = PFNLGeometry()
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
x = torch.from_numpy(synth.uniform_clips(4, 7, 128, 128, seed=1)).cuda()
for prof in (False, True):
    eng.profile(prof)
    for _ in range(3):
        y = eng.forward(x)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        y = eng.forward(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("per-kernel events %s: enqueue %.3f ms per forward, GPU-complete %.3f ms per forward" % ("on " if prof else "off", 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
