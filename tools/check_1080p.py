"""One-off parity check at the 1080p geometry (1x7x270x480 -> 1080x1920) against the fp32 oracle on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
from oracle import pfnl_fast
geom = PFNLGeometry()
w = synth.synthetic_weights(geom, seed=0)
x = np.ascontiguousarray(synth.moving_field_clips(1, 7, 270, 480, seed=3)[0], np.float32)
eng = PFNLEngine(geom); eng.load_weights(w)
y = eng.forward(x)
t0 = time.time()
ref = pfnl_fast.FastOracle(w, 7, 4, 20).forward(x)
print("oracle %.1f s; max|hip - oracle| = %.3e, mean %.3e, PSNR(hip, oracle) = %.1f dB" % (
    time.time() - t0, np.abs(y - ref).max(), np.abs(y - ref).mean(), synth.psnr(y, ref)))
