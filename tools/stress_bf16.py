"""Stress the bf16 kernels: many random geometries (odd sizes, tiles cut by the image edge, 1..4 clips, T in {3,5,7}),
plain and fused 3x3, the 1x1 and the split-operand non-local block, against fp32 torch references on the GPU with the
same bf16-rounded operands (tolerance: one bf16 ulp of the result) and against themselves (bit-exact repeatability)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from pfnl_amd import ops


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def check(out, ref, what):
    d = (out.float() - ref).abs()
    bound = ref.abs() * 2.0 ** -7 + 2e-5
    assert bool((d <= bound).all()), (what, float((d - bound).max()))
    return float((d / (ref.abs() + 1e-3)).max())


def run(seed=0, seconds=60.0, max_iters=None):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    t_end = time.time() + seconds
    n, worst = 0, 0.0
    while time.time() < t_end and (max_iters is None or n < max_iters):
        T = int(rng.choice([3, 5, 7]))
        clips = int(rng.integers(1, 5))
        H, W = int(rng.integers(1, 80)), int(rng.integers(1, 150))
        Fr = clips * T
        x = r16(torch.randn(Fr, H, W, 64, device="cuda"))
        k = (torch.randn(3, 3, 64, 64) * 0.05)
        b = torch.randn(64) * 0.1
        kd = r16(k).cuda().permute(3, 2, 0, 1)
        conv = F.conv2d(x.permute(0, 3, 1, 2), kd, b.cuda(), padding=1).permute(0, 2, 3, 1)
        xb = x.to(torch.bfloat16)
        y = ops.conv3x3_bf16(xb, k.numpy(), b.numpy(), act=True)
        w1 = check(y, F.leaky_relu(conv, 0.2), ("plain", T, clips, H, W))
        assert torch.equal(y, ops.conv3x3_bf16(xb, k.numpy(), b.numpy(), act=True)), ("plain not repeatable", T, clips, H, W)
        add = r16(torch.randn(clips, H, W, 64, device="cuda"))
        res = r16(torch.randn(Fr, H, W, 64, device="cuda"))
        yf = ops.conv3x3_bf16(xb, k.numpy(), b.numpy(), act=True, addend=add.to(torch.bfloat16), add_div=T, resid=res.to(torch.bfloat16))
        w2 = check(yf, F.leaky_relu(conv + add.repeat_interleave(T, 0), 0.2) + res, ("fused", T, clips, H, W))
        k10 = torch.randn(1, 1, 64 * T, 64) * 0.05
        y10 = ops.conv1x1_bf16(xb, k10.numpy(), b.numpy(), act=True, frames_per_item=T)
        cat = x.reshape(clips, T, H, W, 64).permute(0, 2, 3, 1, 4).reshape(clips, H, W, T * 64)
        w3 = check(y10, F.leaky_relu(cat @ r16(k10[0, 0]).cuda() + b.cuda(), 0.2), ("1x1", T, clips, H, W))
        worst = max(worst, w1, w2, w3)
        n += 1
    return n, worst


if __name__ == "__main__":
    n, worst = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("stress_bf16: %d random geometries OK, worst relative error %.2e (one bf16 ulp = 7.8e-3)" % (n, worst))
