"""conv3x3=wsplit (conv_wsplit.hip) against the fp64 spec, next to the direct split-f16 kernel and the f32-MFMA kernel: op level over
plain / fused / ragged geometries, then the whole forward with the option.  python tools/check_wsplit.py"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import pfnl_fast, pfnl_spec  # noqa: E402
from pfnl_amd import ops, synth  # noqa: E402
from pfnl_amd.engine import PFNLEngine  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


bad = 0
for (items, H, W, fused, act) in [(1, 8, 16, False, False), (1, 8, 32, False, True), (7, 10, 38, True, True), (3, 5, 7, False, False), (1, 1, 1, False, True),
                                  (7, 33, 70, True, True), (2, 64, 96, False, True), (28, 24, 40, True, True), (1, 9, 130, False, True),
                                  (21, 16, 32, True, False), (4, 128, 128, False, False), (2, 2, 2, False, False), (1, 30, 18, True, True)]:
    rng = np.random.default_rng(items * 1000 + H * 10 + W)
    x = rng.normal(size=(items, H, W, 64)).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
    kw = {}
    if fused:
        div = 7 if items % 7 == 0 else 1
        add = rng.normal(size=(items // div, H, W, 64)).astype(np.float32)
        res = rng.normal(size=(items, H, W, 64)).astype(np.float32)
        ref = ref + np.repeat(add.astype(np.float64), div, axis=0)
        kw = dict(addend=dev(add), add_div=div, resid=dev(res))
    if act:
        ref = pfnl_spec.lrelu(ref)
    if fused:
        ref = ref + res
    got = ops.conv3x3_winograd(dev(x), k, b, act=act, variant="wsplit", **kw).cpu().numpy()
    s16 = ops.conv3x3_winograd(dev(x), k, b, act=act, variant="split16", **kw).cpu().numpy()
    direct = ops.conv2d(dev(x), k, b, act=act, **kw).cpu().numpy()
    e_w, e_s, e_d = np.abs(got - ref).max(), np.abs(s16 - ref).max(), np.abs(direct - ref).max()
    ok = e_w < 8e-6 * max(1.0, np.abs(ref).max())
    bad += not ok
    print(f"{items}x{H}x{W} fused={fused} act={act}: wsplit {e_w:.3g}  split16 {e_s:.3g}  direct f32 {e_d:.3g}  |ref| {np.abs(ref).max():.2f} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        d = np.abs(got - ref)
        idx = np.unravel_index(np.argmax(d), d.shape)
        print("   worst at", idx, "got", got[idx], "ref", ref[idx], " frac bad:", float((d > 1e-4).mean()))

for (B, T, H, W, nb, scale) in [(1, 7, 16, 24, 2, 4), (2, 7, 32, 32, 20, 4), (1, 5, 20, 36, 3, 2)]:
    geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
    wts = synth.synthetic_weights(geom, seed=0)
    x = synth.uniform_clips(B, T, H, W, seed=5)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(wts)
    eng.set_option("conv3x3", "wsplit")
    y = eng.forward(x)
    ref = pfnl_fast.FastOracle(wts, num_frames=T, scale=scale, num_block=nb).forward(x)
    err = float(np.abs(y - ref).max())
    print(f"forward {B}x{T}x{H}x{W} nb={nb}: max|hip - oracle| = {err:.3g} {'ok' if err < 5e-5 else 'FAIL'}", flush=True)
    bad += not (err < 5e-5)
    eng.close()
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
