"""ms per forward of BATCHES of small clips (default 7x32x32, north_star's second size) under each launch structure the plan could take:
auto, the small-shape trunk forced (small=on), the per-tile launches (small=off, split16_mid=auto), the chained launches (small=off,
split16_mid=off).  Says whether trunk_plan's thresholds (tuned on 128x128-class shapes) pick the fastest structure for tiny images in batches.
usage: python tools/small_batch_sweep.py [H] [W] [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd import synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
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
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        e.forward_device(x.data_ptr(), o.data_ptr(), B, H, W, st)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


VARIANTS = (("auto", {}), ("small", {"small": "on"}), ("per-tile", {"small": "off", "split16_mid": "auto"}), ("chains", {"small": "off", "split16_mid": "off"}))
print("%dx%d %s   B: plan(auto) | " % (H, W, prec) + " | ".join(v[0] for v in VARIANTS) + "   [ms per forward; us per clip of the best]")
for B in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
    row, plans = [], []
    for name, opts in VARIANTS:
        for k, v in opts.items():
            e.set_option(k, v)
        plans.append(e.plan(B, H, W)["structure"])
        row.append(ms(B))
        for k in opts:
            e.set_option(k, "auto")
    best = min(row)
    print("%3d: %-12s | " % (B, plans[0]) + " | ".join("%7.3f%s" % (t, "*" if t == best else " ") for t in row) + "   %6.1f us per clip   (%s)" % (1e3 * best / B, ", ".join(plans[1:])), flush=True)
