"""Random-geometry stress of the round-4 kernels (odd sizes, tiles cut by the image edge, 1..3 clips, T in {3,5,7}):
  * the bf16 3x3 kernels the library selects (conv_bf16_v3.hip for the fused and the conv1_i + conv10_i mode, conv_bf16_v2.hip for the plain one; PFNL_BF16_V3 / PFNL_BF16_V2 select others) through the three modes - plain, fused (addend + residual), conv1_i + conv10_i - against fp32 torch references on
    the GPU with the same bf16-rounded operands (one bf16 ulp), repeatable bit for bit;
  * the two-launch small-shape block of conv_small.hip (conv1_i + conv10_i partials, conv2_i summing them) against the direct f32-MFMA
    kernels (ops.conv2d) on the same data, repeatable bit for bit.
usage: stress_r04.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from pfnl_amd import ops


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def check16(out, ref, what):
    d = (out.float() - ref).abs()
    bound = ref.abs() * 2.0 ** -7 + 2e-5
    assert bool((d <= bound).all()), (what, float((d - bound).max()))


def run(seed=0, seconds=60.0):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    t_end = time.time() + seconds
    n, worst_small = 0, 0.0
    while time.time() < t_end:
        T = int(rng.choice([3, 5, 7])); clips = int(rng.integers(1, 4)); Fr = clips * T
        H, W = int(rng.integers(1, 80)), int(rng.integers(1, 150))
        # ---- bf16, second-generation kernel
        x = r16(torch.randn(Fr, H, W, 64, device="cuda"))
        k = torch.randn(3, 3, 64, 64) * 0.05
        b = torch.randn(64) * 0.1
        conv = F.conv2d(x.permute(0, 3, 1, 2), r16(k).cuda().permute(3, 2, 0, 1), b.cuda(), padding=1).permute(0, 2, 3, 1)
        xb = x.to(torch.bfloat16)
        y = ops.conv3x3_bf16(xb, k.numpy(), b.numpy(), act=True)
        check16(y, F.leaky_relu(conv, 0.2), ("plain", T, clips, H, W))
        assert torch.equal(y, ops.conv3x3_bf16(xb, k.numpy(), b.numpy(), act=True)), ("plain not repeatable", T, clips, H, W)
        add = r16(torch.randn(clips, H, W, 64, device="cuda")); res = r16(torch.randn(Fr, H, W, 64, device="cuda"))
        yf = ops.conv3x3_bf16(xb, k.numpy(), b.numpy(), act=True, addend=add.to(torch.bfloat16), add_div=T, resid=res.to(torch.bfloat16))
        check16(yf, F.leaky_relu(conv + add.repeat_interleave(T, 0), 0.2) + res, ("fused", T, clips, H, W))
        assert torch.equal(yf, ops.conv3x3_bf16(xb, k.numpy(), b.numpy(), act=True, addend=add.to(torch.bfloat16), add_div=T, resid=res.to(torch.bfloat16)))
        k10 = torch.randn(1, 1, 64 * T, 64) * 0.05
        b10 = torch.randn(64) * 0.1
        o1, ob = ops.conv1_conv10_bf16(xb, k.numpy(), b.numpy(), k10.numpy(), b10.numpy(), T)
        check16(o1, F.leaky_relu(conv, 0.2), ("c1c10 inp1", T, clips, H, W))
        cat = o1.float().reshape(clips, T, H, W, 64).permute(0, 2, 3, 1, 4).reshape(clips, H, W, T * 64)     # (the 1x1 takes the ROUNDED inp1)
        check16(ob, F.leaky_relu(cat @ r16(k10[0, 0]).cuda() + b10.cuda(), 0.2), ("c1c10 base", T, clips, H, W))
        p1, pb = ops.conv1_conv10_bf16(xb, k.numpy(), b.numpy(), k10.numpy(), b10.numpy(), T)
        assert torch.equal(o1, p1) and torch.equal(ob, pb), ("c1c10 not repeatable", T, clips, H, W)
        # ---- fp32 small-shape block, two launches
        Hs, Ws = int(rng.integers(1, 70)), int(rng.integers(1, 100))
        xs = torch.from_numpy(rng.normal(size=(Fr, Hs, Ws, 64)).astype(np.float32)).cuda()
        k1 = (rng.normal(size=(3, 3, 64, 64)) / 24.0).astype(np.float32)
        k2 = (rng.normal(size=(3, 3, 128, 64)) / 34.0).astype(np.float32)
        k10f = (rng.normal(size=(1, 1, 64 * T, 64)) / np.sqrt(64 * T)).astype(np.float32)
        b1, b2, bx = ((rng.normal(size=64) * 0.1).astype(np.float32) for _ in range(3))
        g1, g2 = ops.conv_small_pf_block(xs, k1, b1, k10f, bx, k2, b2, T)
        h1, h2 = ops.conv_small_pf_block(xs, k1, b1, k10f, bx, k2, b2, T)
        assert torch.equal(g1, h1) and torch.equal(g2, h2), ("small block not repeatable", T, clips, Hs, Ws)
        r1 = ops.conv2d(xs, k1, b1, act=True)
        rb = ops.conv2d(r1, k10f, bx, act=True, frames_per_item=T)
        r2 = ops.conv2d(r1, np.ascontiguousarray(k2[:, :, 64:]), b2, act=True, addend=ops.conv2d(rb, np.ascontiguousarray(k2[:, :, :64]), None, act=False), add_div=T, resid=xs)
        d = max(float((g1 - r1).abs().max()), float((g2 - r2).abs().max()))
        worst_small = max(worst_small, d)
        assert d < 3e-5, ("small block mismatch", T, clips, Hs, Ws, d)
        n += 1
    return n, worst_small


if __name__ == "__main__":
    n, w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
    print("stress_r04: %d random geometries OK (bf16 3x3 x 3 modes within one bf16 ulp, small two-launch block max |d| %.2e vs the f32-MFMA kernels)" % (n, w))
