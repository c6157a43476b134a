"""PCIe-inclusive timing of the host-pointer path (numpy in -> numpy out, like the reference's
sess.run timing, model/pfnl.py:249-253): H2D + kernels + D2H inside pfnl_forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pfnl_amd import synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
g = PFNLGeometry()
e = PFNLEngine(g); e.load_weights(synth.synthetic_weights(g))
x = synth.uniform_clips(4, 7, 128, 128)
e.forward(x)
ts = []
for _ in range(10):
    t = time.perf_counter(); e.forward(x); ts.append(time.perf_counter() - t)
print("host-pointer path: %.3f ms per 4-clip step (min of 10), %.1f HR frames/s" % (1e3 * min(ts), 4 / min(ts)))
