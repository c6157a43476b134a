"""Random-geometry stress of round 6's launch structures on top of the default forward: batches whose (clip, tile) chains are NOT a whole number
of rounds of the grid (option split16_splitchains: the chains of the partial round cut by frames, conv10_i's partial sums through
c10_finalize_kernel) and the opt-in split16_sf0 (split-format copy of inp0, halo by LDS-DMA).  Every forward is repeated bit for bit; the cut
launch is compared with the uncut one (summation-order noise only) and with strict_fp32 (the f32-MFMA kernels: an independent implementation
of every layer); split16_sf0 must equal the default BIT FOR BIT (same operands, same order).  usage: python tools/stress_r06.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth


worst16 = 0.0


def run(seed=0, seconds=60.0):
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    n, ncut, worst, worst_strict = 0, 0, 0.0, 0.0
    global worst16
    parts = {}
    engines = {}
    while time.time() < t_end:
        T = int(rng.choice([3, 5, 7])); scale = int(rng.choice([2, 4])); nb = int(rng.integers(1, 4))
        key = (T, scale, nb)
        if key not in engines:
            geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
            e = PFNLEngine(geom); e.load_weights(synth.synthetic_weights(geom, seed=int(rng.integers(0, 1000))))
            engines[key] = e
        eng = engines[key]
        # shapes with more than one round of chains: 257 .. ~900 chains
        H, W = 2 * int(rng.integers(8, 80)), 2 * int(rng.integers(16, 100))
        per_clip = ((H + 7) // 8) * ((W + 31) // 32)
        B = int(rng.integers(max(1, 257 // per_clip), max(2, 900 // per_clip) + 1))
        pl = eng.plan(B, H, W)
        if not pl["structure"].startswith("chain2"):
            continue
        x = synth.uniform_clips(B, T, H, W, seed=int(rng.integers(0, 1 << 30)))
        y = eng.forward(x)
        assert np.isfinite(y).all() and np.array_equal(y, eng.forward(x)), ("not repeatable", key, B, H, W, pl)
        eng.set_option("split16_mfma", "32")                        # (split16_sf0's chain kernel is the 32x32x16 one: bit identity is with that kernel ...
        y32 = eng.forward(x)
        eng.set_option("split16_sf0", "on")
        assert np.array_equal(y32, eng.forward(x)), ("split16_sf0 moved a bit", key, B, H, W, pl)
        eng.set_option("split16_sf0", "off")
        eng.set_option("split16_mfma", "16")
        worst16 = max(worst16, float(np.abs(y - y32).max()))        # ... and the default 16x16x32 chain kernel within summation-order noise of it)
        assert np.abs(y - y32).max() < 2e-5, ("16x16x32 chain kernel", key, B, H, W, float(np.abs(y - y32).max()))
        if pl["structure"] == "chain2_split":
            ncut += 1
            parts[(pl["split_parts"], pl["part_frames"])] = parts.get((pl["split_parts"], pl["part_frames"]), 0) + 1
            eng.set_option("split16_splitchains", "off")
            d = float(np.abs(y - eng.forward(x)).max())
            eng.set_option("split16_splitchains", "auto")
            worst = max(worst, d)
            assert d < 2e-5, ("cut vs uncut", key, B, H, W, pl, d)
            eng.set_option("strict_fp32", "on")
            ds = float(np.abs(y - eng.forward(x)).max())
            eng.set_option("strict_fp32", "off")
            worst_strict = max(worst_strict, ds)
            assert ds < 1e-4, ("cut vs strict", key, B, H, W, pl, ds)
        if pl["structure"] == "chain2_split":                         # the bf16 trunk cuts the same chains (conv_bf16_v3.hip SPLIT): repeatable, within bf16
            eng.set_option("precision", "bf16")                      # rounding of its uncut launch
            p16 = eng.plan(B, H, W)
            assert p16["structure"] == "bf16_3_split", p16
            y16 = eng.forward(x)
            assert np.isfinite(y16).all() and np.array_equal(y16, eng.forward(x)), ("bf16 not repeatable", key, B, H, W, p16)
            eng.set_option("split16_splitchains", "off")
            y16u = eng.forward(x)
            eng.set_option("split16_splitchains", "auto")
            eng.set_option("precision", "fp32")
            ps = synth.psnr(y16, y16u)
            assert ps > 55.0 and synth.psnr(y16, y) > 50.0, ("bf16 cut vs uncut", key, B, H, W, ps)
        n += 1
    return n, ncut, worst, worst_strict, parts


if __name__ == "__main__":
    n, ncut, worst, ws, parts = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("stress_r06: %d geometries on the two-launch block (all repeated bit for bit, split16_sf0 bit-equal), %d with split chains "
          "(parts x frames: %s): max |cut - uncut| %.3g, max |cut - strict_fp32| %.3g, max |16x16x32 - 32x32x16 chain kernel| %.3g" % (n, ncut, sorted(parts.items()), worst, ws, worst16))
