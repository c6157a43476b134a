"""Stress the persistent Winograd kernel: many random geometries, repeated calls, every mode, against the per-tile
kernel (same arithmetic -> differences at rounding level) and against itself (bit-exact repeatability)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops


def run(seed=0, seconds=60.0, max_iters=None):
  rng = np.random.default_rng(seed)
  t_end = time.time() + seconds
  n = 0
  worst = 0.0
  while time.time() < t_end and (max_iters is None or n < max_iters):
      T = int(rng.choice([3, 5, 7]))
      clips = int(rng.integers(1, 5))
      H, W = 2 * int(rng.integers(1, 70)), 2 * int(rng.integers(1, 90))
      x = torch.from_numpy(rng.normal(size=(clips * T, H, W, 64)).astype(np.float32)).cuda()
      k = (rng.normal(size=(3, 3, 64, 64)) / 24).astype(np.float32)
      b = rng.normal(size=64).astype(np.float32)
      # mode 0 vs tile kernel
      y_ws = ops.conv3x3_winograd(x, k, b, variant="winograd_ws")
      y_tl = ops.conv3x3_winograd(x, k, b, variant="winograd")
      d0 = (y_ws - y_tl).abs().max().item()
      assert torch.equal(y_ws, ops.conv3x3_winograd(x, k, b, variant="winograd_ws")), ("mode 0 not repeatable", T, clips, H, W)
      # mode 2 (grouped conv2) vs mode-1-style composition with the tile kernel
      base = torch.from_numpy(rng.normal(size=(clips, H, W, 64)).astype(np.float32)).cuda()
      res = torch.from_numpy(rng.normal(size=(clips * T, H, W, 64)).astype(np.float32)).cuda()
      k2 = (rng.normal(size=(3, 3, 128, 64)) / 34).astype(np.float32)
      y_g = ops.conv2_grouped(x, base, k2, b, res, T)
      pb = ops.conv3x3_winograd(base, np.ascontiguousarray(k2[:, :, :64]), None, act=False, variant="winograd")
      y_s = ops.conv3x3_winograd(x, np.ascontiguousarray(k2[:, :, 64:]), b, addend=pb, add_div=T, resid=res, variant="winograd")
      d2 = (y_g - y_s).abs().max().item()
      assert torch.equal(y_g, ops.conv2_grouped(x, base, k2, b, res, T)), ("mode 2 not repeatable", T, clips, H, W)
      # mode 3 (accumulating) vs the direct kernel
      k3 = (rng.normal(size=(3, 3, 64 * T, 48)) / np.sqrt(576 * T)).astype(np.float32)
      b3 = rng.normal(size=48).astype(np.float32)
      y_a = ops.conv3x3_accum(x, k3, b3, frames_per_clip=T)
      y_d = ops.conv2d(x, k3, b3, act=True, frames_per_item=T)
      d3 = (y_a - y_d).abs().max().item()
      assert torch.equal(y_a, ops.conv3x3_accum(x, k3, b3, frames_per_clip=T)), ("mode 3 not repeatable", T, clips, H, W)
      m = max(d0, d2, d3)
      worst = max(worst, m)
      assert m < 3e-5, ("mismatch", T, clips, H, W, d0, d2, d3)
      n += 1
  return n, worst


if __name__ == "__main__":
    n, worst = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("stress ok: %d random geometries, worst |difference| %.2e" % (n, worst))
