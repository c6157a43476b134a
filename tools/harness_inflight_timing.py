"""test_video_lr at the reference harness's REAL shape (model/pfnl.py:264, 332: part = 50 / 1000 -> ONE window per forward for every Vid4 /
UDM10 sequence): a 41-frame 144x180 sequence (Vid4 'city' / 'walk' class), the serial loop (PFNL_HARNESS_INFLIGHT=1) against two forwards in
flight on two handles (= 2, the default for the launch-latency-bound structures).  Reports the device time per frame (the harness's own
'spent ... s in total' over the frames, first batch included), the wall time, and compares the PNG bytes.
usage: python tools/harness_inflight_timing.py [frames] [H] [W] [precision]"""
import contextlib, hashlib, io, os, re, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from model.pfnl import PFNL
from pfnl_amd import synth
from pfnl_amd.spec import PFNLGeometry

F = int(sys.argv[1]) if len(sys.argv) > 1 else 41
H = int(sys.argv[2]) if len(sys.argv) > 2 else 144
W = int(sys.argv[3]) if len(sys.argv) > 3 else 180
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
d = tempfile.mkdtemp()
seq = os.path.join(d, "seq")
os.makedirs(os.path.join(seq, "blur4"))
rng = np.random.default_rng(0)
base = rng.integers(0, 256, size=(H + 2 * F, W + 2 * F, 3), dtype=np.uint8)
for i in range(F):
    Image.fromarray(base[i:i + H, 2 * i:2 * i + W]).save(os.path.join(seq, "blur4", "%04d.png" % i))
m = PFNL()
m.precision = prec
m.save_dir = os.path.join(d, "none")
m.set_weights(synth.synthetic_weights(PFNLGeometry(), seed=0))
print("plan of one window:", m._get_engine().plan(1, H, W))
sha = {}
for mode in ("1", "2", "1", "2"):
    os.environ["PFNL_HARNESS_INFLIGHT"] = mode
    with contextlib.redirect_stdout(io.StringIO()):
        m.test_video_lr(seq, name="warm" + mode, part=50)              # warm-up: allocations, the second handle
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        m.test_video_lr(seq, name="out" + mode, part=50)
    wall = time.perf_counter() - t0
    dev = float(re.search(r"spent ([0-9.eE+-]+) s in total", buf.getvalue()).group(1))
    h = hashlib.sha256()
    for p in sorted(os.listdir(os.path.join(seq, "out" + mode))):
        h.update(np.asarray(Image.open(os.path.join(seq, "out" + mode, p))).tobytes())
    sha[mode] = h.hexdigest()[:16]
    print("%d forward(s) in flight: %d frames %dx%d %s, part=50 (B = 1 per forward): device %.2f ms per frame, wall %.3f s, frames sha %s"
          % (int(mode), F, H, W, prec, 1e3 * dev / F, wall, sha[mode]), flush=True)
print("byte-identical:", sha["1"] == sha["2"])
