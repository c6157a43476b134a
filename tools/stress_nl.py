"""Random-geometry stress of the non-local kernels on the f16 matrix pipe (nonlocal_f16.hip: nl_attn_f16_sw_kernel by default,
the only kernel since round 5) against the fp64 spec: random B, T in {3, 5, 7}, H, W (even), i.e. random key counts across the ring,
tile, half and key-split boundaries, in both operand forms.  A DMA piece that lands late or a ring slot refilled early is a wrong block of
64 keys - far above the tolerances.
usage: python tools/stress_nl.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import pfnl_spec
from pfnl_amd import ops


def run(seed=0, seconds=30.0, max_iters=None):
    rng = np.random.default_rng(seed)
    t_end, n, worst = time.time() + seconds, 0, {"split16": 0.0, "f16": 0.0}
    while time.time() < t_end and (max_iters is None or n < max_iters):
        T = int(rng.choice([3, 5, 7]))
        B = int(rng.integers(1, 4))
        H = 2 * int(rng.integers(1, 41))
        W = 2 * int(rng.integers(1, 61))
        if (H // 2) * (W // 2) > 2600:
            continue
        C = 12 * T
        x = rng.random((B, T, H, W, 3), dtype=np.float32) * float(rng.choice([1.0, 0.3, 0.05]))
        wg = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
        ww = (rng.normal(size=(1, 1, C, C)) / np.sqrt(C)).astype(np.float32)
        bg = rng.normal(size=C).astype(np.float32) * 0.1
        bw = rng.normal(size=C).astype(np.float32) * 0.1
        x64 = x.astype(np.float64)
        stack = np.concatenate([x64[:, t] for t in range(T)], -1)
        z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64),
                                     ww.astype(np.float64), bw.astype(np.float64))
        ref = stack + pfnl_spec.depth_to_space2(z)
        xd = torch.from_numpy(x).cuda()
        for kernel, tol in (("split16", 2e-5), ("f16", 2e-3)):
            got = ops.nonlocal_residual(xd, wg, bg, ww, bw, precision=kernel).cpu().numpy()
            err = float(np.abs(got - ref).max())
            assert err < tol, (kernel, B, T, H, W, err)
            worst[kernel] = max(worst[kernel], err)
        n += 1
    return n, worst


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n, worst = run(seed, secs)
    print("stress_nl: %d random geometries OK, worst |err| vs fp64: %s" % (n, worst))
