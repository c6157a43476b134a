"""Entry script with the shape of the reference's `main.py:1-15`: builds PFNL and runs it.

The reference's default action is ``PFNL().train()`` (out of scope here: inference-only build); the
action it keeps commented out, ``model.testvideos(path)``, is the one wired up.  Usage:
    python main.py /path/to/dataset_root        # <root>/<sequence>/truth/*.png -> <sequence>/pfnl/*.png
"""
import os
import sys

from model.pfnl import PFNL

os.environ.setdefault("HIP_VISIBLE_DEVICES", "0")   # the reference pins CUDA_VISIBLE_DEVICES=0 (main.py:10)

if __name__ == '__main__':
    model = PFNL()
    if len(sys.argv) > 1:
        model.testvideos(sys.argv[1])
    else:
        print("usage: python main.py <dataset_root>   (training, the reference's default action, is out of scope)")
