"""Pins the harness-side numerics against the REFERENCE's own code, run in the build container.

`/root/reference/utils.py` cannot be imported (its first lines import tensorflow, h5py, cv2), but the functions these rows rest on
are pure numpy / scipy: `gkern` (:95-105, the BLUR kernel of the 4x decimation), `_rgb2ycbcr` (:194-211), `to_uint8` (:213-216) and
`AVG_PSNR` (:218-246).  This script parses the file with `ast`, takes exactly those four function definitions out of the tree and
executes THEM - the reference's own statements, compiled from the reference's file at run time - against the real numpy and scipy:
no stand-in for any module, no TF import, nothing of the reference's text is written anywhere.  What is written is data: seeded
inputs and the reference's outputs, to tests/golden/utils_ref.npz.  The GPU box never runs this (no /root/reference there); the
fixture travels.  tests/test_host.py compares synth.gaussian_kernel_1d / blur_decimate, metrics.rgb2ycbcr / to_uint8 / avg_psnr with it.

    python tools/make_utils_golden.py
"""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/utils.py"
WANTED = ("gkern", "_rgb2ycbcr", "to_uint8", "AVG_PSNR")


def reference_functions():
    with open(REF, "rt") as f:
        tree = ast.parse(f.read(), filename=REF)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(d.name for d in defs) == sorted(WANTED), [d.name for d in defs]
    ns = {"np": np}                                            # the one global these functions use (utils.py:2)
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)
    return {n: ns[n] for n in WANTED}


def main():
    ref = reference_functions()
    rng = np.random.default_rng(77)
    blur = np.asarray(ref["gkern"](13, 1.6), np.float64)               # utils.py:104: BLUR = gkern(13, 1.6)
    rgb = rng.integers(0, 256, size=(6, 9, 3)).astype(np.float64)
    ycc255 = ref["_rgb2ycbcr"](rgb.copy(), 255)
    ycc1 = ref["_rgb2ycbcr"](rgb.copy() / 255.0, 1)
    u8_in = np.concatenate([rng.uniform(-0.2, 1.2, size=200), [0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255, 0.0, 1.0]]).astype(np.float32)
    u8_out = ref["to_uint8"](u8_in, 0, 1)
    vid_true = rng.random((7, 28, 24, 3)).astype(np.float32)
    vid_pred = np.clip(vid_true + rng.normal(0, 0.03, size=vid_true.shape), 0, 1).astype(np.float32)
    psnr01 = float(ref["AVG_PSNR"](vid_true, vid_pred, vmin=0, vmax=1))
    psnr255 = float(ref["AVG_PSNR"](vid_true * 255.0, vid_pred * 255.0, vmin=0, vmax=255))
    psnr_b = float(ref["AVG_PSNR"](vid_true, vid_pred, vmin=0, vmax=1, t_border=0, sp_border=4))
    out = os.path.join(ROOT, "tests", "golden", "utils_ref.npz")
    np.savez_compressed(out, blur=blur, rgb=rgb, ycc255=ycc255, ycc1=ycc1, u8_in=u8_in, u8_out=u8_out, vid_true=vid_true,
                        vid_pred=vid_pred, avg_psnr=np.array([psnr01, psnr255, psnr_b]))
    print("wrote", out, "| BLUR centre tap %.7f, sum %.12f | AVG_PSNR %.6f %.6f %.6f" % (blur[6, 6], blur.sum(), psnr01, psnr255, psnr_b))


if __name__ == "__main__":
    main()
