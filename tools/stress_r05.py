"""Random-geometry stress of the DEFAULT forward after round 5's work-order rules (the small-shape trunk below 200 tiles of 8x32 pixels, four
per-tile launches per block below 136 (clip, tile) chains, the two-launch block above): every forward repeated bit for bit, compared with the
same shape under split16_mid=off (summation-order noise only) and under strict_fp32=on (the f32-MFMA kernels: an independent
implementation of every layer); the bf16 path repeated, compared with its chain-order launches (bit-equal with bf16_conv10=separate) and
with the fp32 result.  usage: python tools/stress_r05.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth


def run(seed=0, seconds=60.0):
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    n, worst, worst_strict, worst_bf16 = 0, 0.0, 0.0, 1e9
    kinds = {"small": 0, "mid": 0, "chains": 0}
    engines = {}
    while time.time() < t_end:
        T = int(rng.choice([3, 5, 7])); scale = int(rng.choice([2, 4])); nb = int(rng.integers(1, 4))
        key = (T, scale, nb)
        if key not in engines:
            geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
            e = PFNLEngine(geom); e.load_weights(synth.synthetic_weights(geom, seed=int(rng.integers(0, 1000))))
            engines[key] = e
        eng = engines[key]
        B = int(rng.integers(1, 7)); H, W = 2 * int(rng.integers(1, 100)), 2 * int(rng.integers(1, 120))
        chains = B * ((H + 7) // 8) * ((W + 31) // 32)
        kinds["small" if chains * T < 200 else ("mid" if chains < 136 else "chains")] += 1
        x = synth.uniform_clips(B, T, H, W, seed=int(rng.integers(0, 1 << 30)))
        y = eng.forward(x)
        assert np.isfinite(y).all() and np.array_equal(y, eng.forward(x)), ("not repeatable", key, B, H, W)
        eng.set_option("split16_mid", "off")
        d = float(np.abs(y - eng.forward(x)).max())
        eng.set_option("split16_mid", "auto")
        eng.set_option("strict_fp32", "on")
        ds = float(np.abs(y - eng.forward(x)).max())
        eng.set_option("strict_fp32", "off")
        worst, worst_strict = max(worst, d), max(worst_strict, ds)
        assert d < 2e-5 and ds < 1e-4, ("mismatch", key, B, H, W, d, ds)
        eng.set_option("precision", "bf16")
        y16 = eng.forward(x)
        assert np.isfinite(y16).all() and np.array_equal(y16, eng.forward(x)), ("bf16 not repeatable", key, B, H, W)
        if chains < 136:                                           # the mid structure = the separate launches in flat order: no bit may move
            eng.set_option("split16_mid", "off")
            eng.set_option("bf16_conv10", "separate")
            assert np.array_equal(y16, eng.forward(x)), ("bf16 flat order differs from the chain order", key, B, H, W)
            eng.set_option("bf16_conv10", "fused")
            eng.set_option("split16_mid", "auto")
        eng.set_option("precision", "fp32")
        p16 = synth.psnr(y16, y)
        worst_bf16 = min(worst_bf16, p16)
        assert p16 > 50.0, ("bf16 mismatch", key, B, H, W, p16)
        n += 1
    return n, worst, worst_strict, worst_bf16, kinds


if __name__ == "__main__":
    n, worst, ws, wb, kinds = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("stress_r05 ok: %d random geometries (%s), worst |auto - split16_mid=off| %.2e, worst |auto - strict_fp32| %.2e, worst PSNR(bf16, fp32) %.1f dB"
          % (n, ", ".join("%d %s" % (v, k) for k, v in kinds.items()), worst, ws, wb))
