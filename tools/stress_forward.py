"""Random-geometry stress of the whole forward: default path (persistent kernels, grouped / accumulating modes where
the geometry allows) against the conv3x3=winograd_tile / conv2=split / conv1x1=tiled path and repeatability; then the
bf16 path (precision=bf16, fused and separate conv10) against the fp32 result (PSNR > 50 dB) and its repeatability."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth


def run(seed=0, seconds=60.0):
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    n, worst, worst_bf16 = 0, 0.0, 1e9
    engines = {}
    while time.time() < t_end:
        T = int(rng.choice([3, 5, 7])); scale = int(rng.choice([2, 4])); nb = int(rng.integers(1, 4))
        key = (T, scale, nb)
        if key not in engines:
            geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
            e = PFNLEngine(geom); e.load_weights(synth.synthetic_weights(geom, seed=int(rng.integers(0, 1000))))
            engines[key] = e
        eng = engines[key]
        B = int(rng.integers(1, 5)); H, W = 2 * int(rng.integers(1, 80)), 2 * int(rng.integers(1, 100))
        x = synth.uniform_clips(B, T, H, W, seed=int(rng.integers(0, 1 << 30)))
        for k, v in (("conv3x3", "winograd"), ("conv2", "grouped"), ("conv1x1", "stream")):
            eng.set_option(k, v)
        y = eng.forward(x)
        assert np.array_equal(y, eng.forward(x)), ("not repeatable", key, B, H, W)
        for k, v in (("conv3x3", "winograd_tile"), ("conv2", "split"), ("conv1x1", "tiled")):
            eng.set_option(k, v)
        d = float(np.abs(y - eng.forward(x)).max())
        worst = max(worst, d)
        assert np.isfinite(y).all() and d < 1e-4, ("mismatch", key, B, H, W, d)
        for k, v in (("conv3x3", "winograd"), ("conv2", "grouped"), ("conv1x1", "stream")):
            eng.set_option(k, v)
        eng.set_option("precision", "bf16")
        y16 = eng.forward(x)
        assert np.array_equal(y16, eng.forward(x)), ("bf16 not repeatable", key, B, H, W)
        eng.set_option("bf16_conv10", "separate")
        y16s = eng.forward(x)
        eng.set_option("bf16_conv10", "fused")
        eng.set_option("precision", "fp32")
        p16, p16s = synth.psnr(y16, y), synth.psnr(y16s, y)
        worst_bf16 = min(worst_bf16, p16, p16s)
        assert np.isfinite(y16).all() and p16 > 50.0 and p16s > 50.0, ("bf16 mismatch", key, B, H, W, p16, p16s)
        n += 1
    return n, worst, worst_bf16


if __name__ == "__main__":
    n, worst, worst_bf16 = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("forward stress ok: %d random geometries, worst |default - tile/split/tiled| %.2e, worst PSNR(bf16, fp32) %.1f dB" % (n, worst, worst_bf16))
