"""Strided-subsample goldens of the FULL-SIZE oracle outputs for BASELINE.json configs[3] (1x7x270x480 -> 1080x1920), so
that the 1080p forward is checked against the ORACLE inside `pytest -m gpu` (the oracle needs minutes at this size on a
small host; on the GPU box only the committed subsample travels).

    python tools/make_golden_1080p.py        # ~5 min on 8 cores

Writes tests/golden/cfg4_1080p_stride8.npz: the oracle output at every 8th HR pixel ([135,240,3] float32) for
  y_fp32      - oracle/pfnl_fast.py, fp32 graph (also what the fp32 build is compared with)
  y_bf16mode  - the same with trunk_dtype="bf16" (rounding points of the bf16 trunk)
plus a 64x64 dense HR crop of each (rows/cols 512..575 / 960..1023).  Input = synth.uniform_clips(1,7,270,480, seed 4040),
weights = synth.synthetic_weights(seed 0): both regenerated from the seeds by the test.
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

SEED, STRIDE = 4040, 8
CROP = (slice(512, 576), slice(960, 1024))


def main():
    geom = PFNLGeometry()
    w = synth.synthetic_weights(geom, seed=0)
    x = synth.uniform_clips(1, 7, 270, 480, seed=SEED)
    rec = {"meta": np.array([SEED, STRIDE, CROP[0].start, CROP[1].start, 64], np.int64)}
    for key, dt in (("y_fp32", "fp32"), ("y_bf16mode", "bf16")):
        t0 = time.time()
        y = pfnl_fast.FastOracle(w, trunk_dtype=dt).forward(x)[0, 0]          # [1080,1920,3]
        rec[key] = y[::STRIDE, ::STRIDE].astype(np.float32)
        rec[key + "_crop"] = y[CROP].astype(np.float32)
        print(key, "%.1f s" % (time.time() - t0), y.shape, float(y.mean()))
    out = os.path.join(ROOT, "tests", "golden", "cfg4_1080p_stride8.npz")
    np.savez_compressed(out, **rec)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
