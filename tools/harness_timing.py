"""test_video_lr on a 32-frame 270x480 sequence (-> 1080p): share of the wall time spent outside pfnl_forward (PNG decode,
H2D of the LR frames - once -, window gather, uint8 quantisation, D2H, PNG encode on worker threads)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from model.pfnl import PFNL
from pfnl_amd import synth
from pfnl_amd.spec import PFNLGeometry
d = tempfile.mkdtemp()
seq = os.path.join(d, "seq")
os.makedirs(os.path.join(seq, "blur4"))
rng = np.random.default_rng(0)
base = rng.integers(0, 256, size=(270 + 64, 480 + 64, 3), dtype=np.uint8)
for i in range(32):
    Image.fromarray(base[i:i + 270, 2 * i:2 * i + 480]).save(os.path.join(seq, "blur4", "%04d.png" % i))
m = PFNL()
m.save_dir = os.path.join(d, "none")
m.set_weights(synth.synthetic_weights(PFNLGeometry(), seed=0))
import contextlib, io, re
m._get_engine()
m.test_video_lr(seq, name="warm", part=32)                      # warm-up: allocations, first-call costs
for part in (8, 32):
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        m.test_video_lr(seq, name="out%d" % part, part=part)
    wall = time.perf_counter() - t0
    dev = float(re.search(r"spent ([0-9.eE+-]+) s in total", buf.getvalue()).group(1))
    print("part=%d: wall %.3f s, device time of the batches (gather + pfnl_forward + quantise + D2H) %.3f s -> %.1f %% of the wall outside" % (
        part, wall, dev, 100 * (1 - dev / wall)))
