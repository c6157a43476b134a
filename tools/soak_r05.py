"""Round-5 soak: run-to-run determinism of every shipped forward configuration on the FINAL kernels (the round's inline-asm changes:
conv_split16.hip / conv_sf.hip scalar multiplies, spread operand reads and wave priorities; conv_wsplit.hip, whose MFMAs, transforms and
waits are all inline asm; the clip-chunked non-local launch).  A result read before the matrix pipe has written it, or an LDS slot refilled
early, shows up as a difference between two runs of the same input (that is how the ASAN build found the round's one hazard, DESIGN.md
R5.1): every forward is repeated and compared bit for bit with its first run, and configurations that must agree bit for bit with each
other are covered by the tests (eager against hipGraph replay; the non-local launch in clip chunks: tests/test_gpu_ops.py).
usage: python tools/soak_r05.py [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry

# (label, geometry kwargs, options, B, H, W)
CASES = [
    ("fp32 configs[1]", {}, {}, 4, 128, 128),
    ("fp32 configs[0]", {}, {}, 1, 32, 32),
    ("fp32 configs[4] 2x T=5", {"scale": 2, "num_frames": 5}, {}, 1, 64, 64),
    ("fp32 Vid4 144x180", {}, {}, 2, 144, 180),
    ("fp32 ragged 66x130", {}, {}, 3, 66, 130),
    ("bf16 1080p", {}, {"precision": "bf16"}, 1, 270, 480),
    ("fp32 1080p", {}, {}, 1, 270, 480),
    ("strict fp32 128x128", {}, {"strict_fp32": "on"}, 1, 128, 128),
]


def run(label, gk, opts, B, H, W, reps):
    g = PFNLGeometry(**gk)
    e = PFNLEngine(g)
    e.load_weights(synth.synthetic_weights(g, seed=0))
    for k, v in opts.items():
        e.set_option(k, v)
    x = torch.from_numpy(synth.uniform_clips(B, g.num_frames, H, W, seed=B + H)).cuda()
    first = None
    for r in range(reps):
        y = e.forward(x).cpu().numpy()
        if first is None:
            first = y.copy()
        elif not np.array_equal(first, y):
            raise SystemExit("NONDETERMINISTIC: %s run %d differs from run 0 in %d values (max %g)"
                             % (label, r, int((first != y).sum()), float(np.abs(first - y).max())))
    e.close()
    return first


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    for c in CASES:
        y = run(*c, reps)
        print("%s (%dx%dx%dx%d): %d runs bit-identical" % (c[0], c[3], PFNLGeometry(**c[1]).num_frames, c[4], c[5], reps), flush=True)
    print("soak_r05: ok")
