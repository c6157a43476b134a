"""Oracle goldens at the geometries the reference's own harness exists for (reference model/pfnl.py:25, 86, 203-332):
Vid4 LR 144x180 (calendar), 144x176 (city), 120x180 (foliage / walk), UDM10 LR 180x318 (W not a multiple of 32, N = 14 310 a multiple
of 2 only), and the reference's eval batch [4,7,128,240,3].  The data sets are not in the image: inputs are seeded synthetic clips of
those SHAPES (synth.uniform_clips), weights synth.synthetic_weights(seed 0), 20 blocks.  Stored per geometry (arrays only): every 8th
HR pixel of oracle/pfnl_fast.py's fp32 output and a dense 48x48 HR crop at an M-block / tile boundary.

    python tools/make_golden_ref_geoms.py      # ~1 min on 8 cores -> tests/golden/ref_geoms_stride8.npz
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pfnl_fast  # noqa: E402
from pfnl_amd import synth  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402

STRIDE, CROP = 8, 48
GEOMS = {            # name: (B, H, W, seed)
    "vid4_144x180": (1, 144, 180, 5101),
    "vid4_144x176": (1, 144, 176, 5102),
    "vid4_120x180": (1, 120, 180, 5103),
    "udm10_180x318": (1, 180, 318, 5104),
    "eval_4x128x240": (4, 128, 240, 5105),
}


def crop_origin(H, W):
    """HR origin of the dense crop: across the LR tile corner (8 k, 32 m) nearest the middle, so that tile seams are inside it."""
    y0 = 4 * (8 * ((H // 2) // 8)) - CROP // 2
    x0 = 4 * (32 * ((W // 2) // 32)) - CROP // 2
    return max(0, min(y0, 4 * H - CROP)), max(0, min(x0, 4 * W - CROP))


def main():
    geom = PFNLGeometry()
    w = synth.synthetic_weights(geom, seed=0)
    fo = pfnl_fast.FastOracle(w)
    rec = {"meta": np.array([STRIDE, CROP], np.int64)}
    for name, (B, H, W, seed) in GEOMS.items():
        t0 = time.time()
        x = synth.uniform_clips(B, 7, H, W, seed=seed)
        y = fo.forward(x)[:, 0]                                       # [B, 4H, 4W, 3]
        cy, cx = crop_origin(H, W)
        rec[name + "_shape"] = np.array([B, H, W, seed, cy, cx], np.int64)
        rec[name + "_sub"] = y[:, ::STRIDE, ::STRIDE].astype(np.float32)
        rec[name + "_crop"] = y[:, cy:cy + CROP, cx:cx + CROP].astype(np.float32)
        print(name, "%.1f s" % (time.time() - t0), y.shape, float(y.mean()), flush=True)
    out = os.path.join(ROOT, "tests", "golden", "ref_geoms_stride8.npz")
    np.savez_compressed(out, **rec)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
