"""Generates tests/golden/*.npz with the fp64 spec oracle (oracle/pfnl_spec.py) in the build
container.  The reference itself cannot run here (TensorFlow 1.12 is not installable, SURVEY.md
§8(c)), so these vectors pin the build against its own spec restatement ("parity unpinned").

    python tools/make_golden.py

Each file holds the seeded inputs (float32), the weight seed/geometry needed to regenerate the
weights with pfnl_amd.synth.synthetic_weights, and the expected outputs (float32 of the fp64 result).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pfnl_spec  # noqa: E402
from pfnl_amd import synth  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CASES = [
    # name, T, scale, num_block, B, H, W, input kind, seed
    ("cfg1_7x32x32", 7, 4, 20, 1, 32, 32, "uniform", 1234),          # BASELINE.json configs[0]
    ("field_7x32x32", 7, 4, 20, 1, 32, 32, "field", 77),              # smooth clip + GT (PSNR parity)
    ("ragged_7x20x36_nb2", 7, 4, 2, 2, 20, 36, "uniform", 5),         # N=180: MFMA tile edges, W%32!=0
    ("x2_5x16x24_nb2", 5, 2, 2, 1, 16, 24, "uniform", 6),             # configs[4]-style geometry
    ("t3_3x12x40_nb1", 3, 4, 1, 1, 12, 40, "uniform", 7),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, T, scale, nb, B, H, W, kind, seed in CASES:
        geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
        w = synth.synthetic_weights(geom, seed=0)
        gt = None
        if kind == "uniform":
            x = synth.uniform_clips(B, T, H, W, seed=seed)
        else:
            x, gt = synth.moving_field_clips(B, T, H, W, scale=scale, seed=seed)
        taps = {}
        y = pfnl_spec.forward(x, w, scale=scale, num_block=nb, taps=taps)
        rec = dict(x=x.astype(np.float32), y=y.astype(np.float32),
                   nl_out=taps["nl_out"].astype(np.float32), merge1=taps["merge1"].astype(np.float32),
                   bic=taps["bic"].astype(np.float32),
                   meta=np.array([T, scale, nb, B, H, W, 0], np.int64))
        if gt is not None:
            rec["gt"] = gt.astype(np.float32)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
        print(name, y.shape, float(np.abs(y).max()))

    # harness fixture: 5-frame 16x24 LR sequence -> windows -> SR uint8 (model/pfnl.py:236-258)
    geom = PFNLGeometry(num_block=1)
    w = synth.synthetic_weights(geom, seed=0)
    rng = np.random.default_rng(99)
    lr_u8 = rng.integers(0, 256, size=(5, 16, 24, 3), dtype=np.uint8)
    lrs = (lr_u8 / 255.).astype(np.float32)
    win = pfnl_spec.sliding_windows(lrs, 7)
    sr = pfnl_spec.forward(win, w, num_block=1)
    np.savez_compressed(os.path.join(OUT, "harness_5x16x24_nb1.npz"), lr_u8=lr_u8,
                        sr_u8=pfnl_spec.quantise(sr[:, 0]), sr=sr.astype(np.float32))
    print("harness", sr.shape)

    # bf16 path (BASELINE.json configs[3]'s arithmetic, build-defined): outputs of the fast oracle's trunk_dtype="bf16"
    # mode for two existing cases (inputs are the cases' x; only y is stored)
    from oracle import pfnl_fast
    rec = {}
    for name, T, scale, nb, B, H, W, kind, seed in CASES:
        if name not in ("ragged_7x20x36_nb2", "x2_5x16x24_nb2"):
            continue
        geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
        w = synth.synthetic_weights(geom, seed=0)
        x = np.load(os.path.join(OUT, name + ".npz"))["x"]
        rec[name] = pfnl_fast.FastOracle(w, T, scale, nb, trunk_dtype="bf16").forward(x).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "bf16_outputs.npz"), **rec)
    print("bf16", {k: v.shape for k, v in rec.items()})


if __name__ == "__main__":
    main()
