"""Pins pfnl_amd.metrics.ssim against the REFERENCE's own implementation, run in the build container.

`/root/reference/modules/SSIM_Index.py` (compute_ssim) needs only numpy + scipy, so unlike the TF1 graph it can be
imported here.  The one incompatibility is the `np.float` alias removed in numpy 1.24; it is restored below before the
import (an alias of a numpy name, not a stand-in for reference code).  The script imports the reference at run time and
writes inputs + the reference's outputs to tests/golden/ssim_ref.npz - data only; nothing of the reference's source is
stored.  The GPU box never runs this (the reference does not exist there); the committed fixture travels.

    python tools/make_metric_golden.py
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/modules/SSIM_Index.py"


def main():
    if not hasattr(np, "float"):
        np.float = float                                   # numpy < 1.24 alias used at SSIM_Index.py:36-37
    spec = importlib.util.spec_from_file_location("ref_ssim_index", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(2024)
    cases_a, cases_b, vals = [], [], []
    for k, (h, w) in enumerate([(32, 48), (64, 64), (17, 23), (40, 12)]):
        a = rng.integers(0, 256, size=(h, w)).astype(np.float64)
        if k == 0:
            b = a.copy()                                       # identical images: SSIM = 1
        elif k == 1:
            b = np.clip(a + rng.normal(0, 12.0, size=a.shape), 0, 255)
        elif k == 2:
            b = rng.integers(0, 256, size=(h, w)).astype(np.float64)     # unrelated
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            a = 127.5 + 100.0 * np.sin(0.3 * yy) * np.cos(0.2 * xx)      # smooth pair
            b = 127.5 + 100.0 * np.sin(0.3 * yy + 0.2) * np.cos(0.2 * xx)
        pad_a = np.zeros((64, 64)); pad_b = np.zeros((64, 64))
        pad_a[:h, :w] = a; pad_b[:h, :w] = b
        cases_a.append(pad_a); cases_b.append(pad_b)
        vals.append([h, w, float(mod.compute_ssim(a, b)), float(mod.compute_ssim(a / 255.0, b / 255.0, l=1.0))])
    out = os.path.join(ROOT, "tests", "golden", "ssim_ref.npz")
    np.savez_compressed(out, a=np.stack(cases_a), b=np.stack(cases_b), hw_ssim255_ssim1=np.array(vals))
    print("wrote", out, np.array(vals)[:, 2:])


if __name__ == "__main__":
    main()
