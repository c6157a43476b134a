"""Where does the fused split16 kernel differ from the fp64 oracle?  (debug aid)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pfnl_spec
from pfnl_amd import ops
rng = np.random.default_rng(0)
items, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (21, 16, 32)
div = 7 if items % 7 == 0 else 1
x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
b = (rng.normal(size=64) * 0.1).astype(np.float32)
add = rng.normal(size=(items // div, H, W, 64)).astype(np.float32)
res = rng.normal(size=(items, H, W, 64)).astype(np.float32)
conv = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
ref = conv + np.repeat(add.astype(np.float64), div, axis=0) + res
d = lambda a: torch.from_numpy(a).cuda()
for rep in range(2):
    got = ops.conv3x3_winograd(d(x), k, b, act=False, variant="split16", addend=d(add), add_div=div, resid=d(res)).cpu().numpy()
    e = np.abs(got - ref)
    bad = e > 1e-4
    print("rep", rep, "max err %.3g, bad elements %d of %d" % (e.max(), bad.sum(), bad.size))
    print("  bad per item ", bad.sum(axis=(1, 2, 3)))
    print("  bad per row  ", bad.sum(axis=(0, 2, 3)))
    print("  bad per col  ", bad.sum(axis=(0, 1, 3)))
    print("  bad per ch   ", bad.sum(axis=(0, 1, 2)))
    if bad.any():
        i, y, xx, c = np.argwhere(bad)[0]
        print("  first bad", (i, y, xx, c), "got", got[i, y, xx, c], "ref", ref[i, y, xx, c], "conv", conv[i, y, xx, c], "add", add[i // div, y, xx, c], "res", res[i, y, xx, c])
        # does got match the reference with another item's addend / residual ?
        for j in range(items):
            alt = conv[i, y, xx, c] + add[i // div, y, xx, c] + res[j, y, xx, c]
            if abs(alt - got[i, y, xx, c]) < 1e-4: print("    = ref with resid of item", j)
        for j in range(items // div):
            alt = conv[i, y, xx, c] + add[j, y, xx, c] + res[i, y, xx, c]
            if abs(alt - got[i, y, xx, c]) < 1e-4: print("    = ref with addend of item", j)
        for dy in range(-2, 3):
            yy = y + dy
            if 0 <= yy < H:
                for nm, arr in (("res", res[i]), ("add", add[i // div])):
                    alt = conv[i, y, xx, c] + (add[i // div, y, xx, c] if nm == "res" else res[i, y, xx, c]) + arr[yy, xx, c]
                    if dy and abs(alt - got[i, y, xx, c]) < 1e-4: print("    = ref with %s taken from row %+d" % (nm, dy))
