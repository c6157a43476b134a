"""Random-geometry stress of the round-2 default path (conv3x3=split16 forced, conv1x1=split16, nonlocal=split16) against the
round-1 f32-MFMA path (winograd / stream / f32) and repeatability; plus the op hook against the direct kernel on random shapes
(chains of T frames, ragged tiles, single rows / columns)."""
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
        # --- op level
        T = int(rng.choice([1, 3, 5, 7])); clips = int(rng.integers(1, 4)); items = clips * T
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 140))
        fused = bool(rng.integers(0, 2)) and T > 1
        x = torch.from_numpy(rng.normal(size=(items, H, W, 64)).astype(np.float32)).cuda()
        k = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
        b = (rng.normal(size=64) * 0.1).astype(np.float32)
        kw = {}
        if fused:
            kw = dict(addend=torch.from_numpy(rng.normal(size=(clips, H, W, 64)).astype(np.float32)).cuda(), add_div=T,
                      resid=torch.from_numpy(rng.normal(size=(items, H, W, 64)).astype(np.float32)).cuda())
        act = bool(rng.integers(0, 2))
        a = ops.conv3x3_winograd(x, k, b, act=act, variant="split16", **kw)
        assert torch.equal(a, ops.conv3x3_winograd(x, k, b, act=act, variant="split16", **kw)), ("op not repeatable", items, H, W, fused)
        d = float((a - ops.conv2d(x, k, b, act=act, **kw)).abs().max())
        worst_op = max(worst_op, d)
        assert d < 2e-5, ("op mismatch", items, T, H, W, fused, act, d)
        # --- whole forward
        T = int(rng.choice([3, 5, 7])); scale = int(rng.choice([2, 4])); nb = int(rng.integers(1, 4))
        key = (T, scale, nb)
        if key not in engines:
            geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
            e = PFNLEngine(geom); e.load_weights(synth.synthetic_weights(geom, seed=int(rng.integers(0, 1000))))
            engines[key] = e
        eng = engines[key]
        B = int(rng.integers(1, 5)); H, W = 2 * int(rng.integers(1, 80)), 2 * int(rng.integers(1, 100))
        x = synth.uniform_clips(B, T, H, W, seed=int(rng.integers(0, 1 << 30)))
        for kk, v in (("conv3x3", "split16"), ("conv1x1", "split16"), ("nonlocal", "split16")):
            eng.set_option(kk, v)
        y = eng.forward(x)
        assert np.array_equal(y, eng.forward(x)), ("not repeatable", key, B, H, W)
        for kk, v in (("conv3x3", "winograd"), ("conv1x1", "stream"), ("nonlocal", "f32")):
            eng.set_option(kk, v)
        d = float(np.abs(y - eng.forward(x)).max())
        worst = max(worst, d)
        assert np.isfinite(y).all() and d < 2e-5, ("mismatch", key, B, H, W, d)
        n += 1
    return n, worst, worst_op


if __name__ == "__main__":
    n, worst, worst_op = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("split16 stress ok: %d random geometries; worst |split16 path - f32-MFMA path| %.2e (forward), %.2e (3x3 op vs direct)" % (n, worst, worst_op))
