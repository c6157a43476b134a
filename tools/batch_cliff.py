"""ms per forward against the batch size at 7x128x128 (64 (clip, tile) chains per clip, 256 workgroups): the cliff behind every whole round of
chains, with the chains of a partial round cut by frames (option split16_splitchains=auto, the default) and uncut (off).  DESIGN.md R6.4.
usage: python tools/batch_cliff.py [H] [W] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd import synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry

H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
g = PFNLGeometry()
e = PFNLEngine(g)
e.load_weights(synth.synthetic_weights(g, seed=0))
e.set_option("precision", prec)
st = torch.cuda.current_stream().cuda_stream


def ms(B):
    x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=B)).cuda()
    o = torch.empty(e.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    for _ in range(5):
        e.forward_device(x.data_ptr(), o.data_ptr(), B, H, W, st)
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        e.forward_device(x.data_ptr(), o.data_ptr(), B, H, W, st)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:                               # clock ramp
    ms(4)
print("B   plan (auto)      ms auto   ms off   auto/B=4   off/B=4   work/B=4")
base = None
for B in (3, 4, 5, 6, 7, 8, 9, 10, 12):
    e.set_option("split16_splitchains", "auto")
    pl = e.plan(B, H, W)
    a = ms(B)
    e.set_option("split16_splitchains", "off")
    b = ms(B)
    if B == 4:
        base = (a, b)
    print("%-3d %-16s %7.3f  %7.3f   %s" % (B, pl["structure"] + ("(%d x %d)" % (pl["split_parts"], pl["part_frames"]) if pl.get("split_parts") else ""), a, b,
                                            ("%.2f       %.2f      %.2f" % (a / base[0], b / base[1], B / 4.0)) if base else ""), flush=True)
