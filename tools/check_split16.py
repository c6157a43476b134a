"""GPU check of conv3x3_split16 (conv_split16.hip): op parity against the fp64 oracle on awkward geometries, error next to
the direct f32-MFMA and Winograd kernels, then whole-forward parity and timing with conv3x3=split16."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pfnl_fast, pfnl_spec  # noqa: E402
from pfnl_amd import ops, synth  # noqa: E402
from pfnl_amd.engine import PFNLEngine  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def case(rng, items, H, W, fused, act, scale=1.0):
    x = (rng.normal(size=(items, H, W, 64)) * scale).astype(np.float32)
    k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
    b = (rng.normal(size=64) * 0.1).astype(np.float32)
    ref = pfnl_spec.conv2d_same(x.astype(np.float64), k.astype(np.float64), b.astype(np.float64))
    kw = {}
    res = None
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
    out = {}
    for variant in ("split16", "winograd_ws") if (H % 2 == 0 and W % 2 == 0) else ("split16",):
        got = ops.conv3x3_winograd(dev(x), k, b, act=act, variant=variant, **kw).cpu().numpy()
        out[variant] = float(np.abs(got - ref).max())
    got = ops.conv2d(dev(x), k, b, act=act, **kw).cpu().numpy()
    out["direct"] = float(np.abs(got - ref).max())
    return out, float(np.abs(ref).max())


def main():
    rng = np.random.default_rng(0)
    bad = 0
    for items, H, W, fused, act in [(1, 8, 32, False, True), (2, 16, 64, False, True), (7, 10, 38, True, True), (14, 20, 36, True, True),
                                    (3, 5, 7, False, False), (1, 1, 1, False, True), (7, 33, 70, True, True), (2, 64, 96, False, True),
                                    (28, 24, 40, True, True), (1, 9, 130, False, True), (9, 48, 48, False, True), (21, 16, 32, True, False)]:
        e, mag = case(rng, items, H, W, fused, act)
        ok = e["split16"] < 4e-6 * max(1.0, mag)
        bad += not ok
        print("items %2d %3dx%3d fused %d act %d  max|ref| %.2f  err: %s %s" % (items, H, W, fused, act, mag, e, "" if ok else "  <-- FAIL"))
    e, mag = case(rng, 2, 16, 32, False, True, scale=1000.0)
    print("large activations (x1000): ", e, mag)
    e, mag = case(rng, 2, 16, 32, False, True, scale=1e-3)
    print("small activations (x1e-3): ", e, mag)
    # whole forward
    for geom, B, H, W in ((PFNLGeometry(num_block=2), 2, 20, 36), (PFNLGeometry(), 1, 32, 32), (PFNLGeometry(num_frames=5, scale=2, num_block=3), 1, 16, 24)):
        w = synth.synthetic_weights(geom, seed=0)
        x = synth.uniform_clips(B, geom.num_frames, H, W, seed=3)
        eng = PFNLEngine(geom, device=0)
        eng.load_weights(w)
        ref = pfnl_spec.forward(x, w, scale=geom.scale, num_block=geom.num_block)
        yw = eng.forward(x)
        eng.set_option("conv3x3", "split16")
        ys = eng.forward(x)
        eng.set_option("conv3x3", "direct")
        yd = eng.forward(x)
        print("forward T%d s%d nb%d %dx%dx%d: max|err| vs fp64 spec: winograd %.3g  split16 %.3g  direct %.3g" % (
            geom.num_frames, geom.scale, geom.num_block, B, H, W, np.abs(yw - ref).max(), np.abs(ys - ref).max(), np.abs(yd - ref).max()))
        bad += not (np.abs(ys - ref).max() < 5e-5)
        eng.close()
    # full size vs fast oracle + timing
    geom = PFNLGeometry()
    w = synth.synthetic_weights(geom, seed=0)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(w)
    x = synth.uniform_clips(4, 7, 128, 128, seed=8)
    torch.set_num_threads(16)
    ref = pfnl_fast.FastOracle(w).forward(x[:1])
    for algo in ("winograd", "split16", "direct"):
        eng.set_option("conv3x3", algo)
        y = eng.forward(x)
        xd = torch.from_numpy(x).cuda()
        out = torch.empty(eng.out_shape(4, 128, 128), dtype=torch.float32, device="cuda")
        for _ in range(3):
            eng.forward_device(xd.data_ptr(), out.data_ptr(), 4, 128, 128, 0)
        torch.cuda.synchronize()
        eng.profile_reset()
        eng.profile(1)
        t0 = time.perf_counter()
        for _ in range(10):
            eng.forward_device(xd.data_ptr(), out.data_ptr(), 4, 128, 128, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        eng.profile(0)
        pr = eng.profile_read()
        print("cfg2 %-9s: %.3f ms/step (%.0f fps)  max|err| vs fp32 oracle (clip 0) %.3g   per class ms: %s" % (
            algo, dt * 1e3, 4 / dt, np.abs(y[:1] - ref).max(), {k: round(v["ms"] / 10, 3) for k, v in pr.items()}))
    print("FAILURES:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
