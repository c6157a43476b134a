"""Random-geometry stress of the round-3 kernels: the split-format chain (conv1_i writing SF -> conv10_i on SF -> conv2_i as one
chain launch: `conv3x3=split16` forced so that every shape takes it), the small-shape trunk (`small=on`) and the default (auto)
choice, each against the f32-MFMA path (`strict_fp32=on`) on the same engine; the chain / SF / small op hooks against the direct f32
kernel; bit-exact repeatability of every call.  usage: stress_r03.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import ops, synth


def run(seed=0, seconds=60.0):
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    n, worst, worst_op = 0, 0.0, 0.0
    engines = {}
    while time.time() < t_end:
        # --- op level: the chain kernel and the small-shape kernel on the same conv2_i problem, against the direct f32 kernel
        T = int(rng.choice([3, 5, 7])); clips = int(rng.integers(1, 4)); F = clips * T
        H, W = int(rng.integers(1, 60)), int(rng.integers(1, 140))
        x = torch.from_numpy(rng.normal(size=(F, H, W, 64)).astype(np.float32)).cuda()
        base = torch.from_numpy(rng.normal(size=(clips, H, W, 64)).astype(np.float32)).cuda()
        res = torch.from_numpy(rng.normal(size=(F, H, W, 64)).astype(np.float32)).cuda()
        k2 = (rng.normal(size=(3, 3, 128, 64)) / 34.0).astype(np.float32)
        b = (rng.normal(size=64) * 0.1).astype(np.float32)
        chain = ops.conv3x3_winograd(x, k2, b, act=True, addend=base, add_div=T, resid=res, variant="split16_sf_chain")
        assert torch.equal(chain, ops.conv3x3_winograd(x, k2, b, act=True, addend=base, add_div=T, resid=res, variant="split16_sf_chain")), ("chain not repeatable", T, clips, H, W)
        small = ops.conv_small(x, k2, b, a=base, a_div=T, resid=res)
        # ... and the fused conv1_i + conv10_i launch against the direct f32 kernels, repeatable bit for bit
        k1 = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
        k10 = (rng.normal(size=(1, 1, 64 * T, 64)) / np.sqrt(64 * T)).astype(np.float32)
        f1, fb = ops.conv1_conv10_split16(x, k1, b, k10, b, T)
        g1, gb = ops.conv1_conv10_split16(x, k1, b, k10, b, T)
        assert torch.equal(f1, g1) and torch.equal(fb, gb), ("c1c10 not repeatable", T, clips, H, W)
        r1 = ops.conv2d(x, k1, b, act=True)
        rb = ops.conv2d(r1, k10, b, act=True, frames_per_item=T)
        d10 = max(float((f1 - r1).abs().max()), float((fb - rb).abs().max()))
        worst_op = max(worst_op, d10)
        assert d10 < 2e-5, ("c1c10 mismatch", T, clips, H, W, d10)
        pb = ops.conv2d(base, np.ascontiguousarray(k2[:, :, :64]), None, act=False)
        ref = ops.conv2d(x, np.ascontiguousarray(k2[:, :, 64:]), b, act=True, addend=pb, add_div=T, resid=res)
        d = max(float((chain - ref).abs().max()), float((small - ref).abs().max()))
        worst_op = max(worst_op, d)
        assert d < 2e-5, ("op mismatch", T, clips, H, W, d)
        # --- whole forward
        T = int(rng.choice([3, 5, 7])); scale = int(rng.choice([2, 4])); nb = int(rng.integers(1, 4))
        key = (T, scale, nb)
        if key not in engines:
            geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
            e = PFNLEngine(geom); e.load_weights(synth.synthetic_weights(geom, seed=int(rng.integers(0, 1000))))
            engines[key] = e
        eng = engines[key]
        B = int(rng.integers(1, 5)); H, W = 2 * int(rng.integers(1, 70)), 2 * int(rng.integers(1, 90))
        xs = synth.uniform_clips(B, T, H, W, seed=int(rng.integers(0, 1 << 30)))
        eng.set_option("strict_fp32", "on")
        yref = eng.forward(xs)
        eng.set_option("strict_fp32", "off")
        for opts in ({"conv3x3": "split16", "small": "off"}, {"conv3x3": "auto", "small": "on"}, {"conv3x3": "auto", "small": "auto"}):
            for kk, v in opts.items():
                eng.set_option(kk, v)
            y = eng.forward(xs)
            assert np.array_equal(y, eng.forward(xs)), ("not repeatable", key, B, H, W, opts)
            d = float(np.abs(y - yref).max())
            worst = max(worst, d)
            assert np.isfinite(y).all() and d < 2e-5, ("mismatch", key, B, H, W, opts, d)
        assert eng.range_reruns() == 0
        n += 1
    return n, worst, worst_op


if __name__ == "__main__":
    n, worst, worst_op = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("round-3 stress ok: %d random geometries; worst |f16-pipe paths - f32-MFMA path| %.2e (forward), %.2e (conv2_i ops vs direct)" % (n, worst, worst_op))
