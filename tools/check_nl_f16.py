"""Non-local block of precision=bf16: split-bf16 operands (nonlocal_bf16.hip) vs plain f16 operands (nonlocal_f16.hip, hi parts only):
error of the op against the fp64 spec, then the 1080p bf16 forward with either (time, PSNR against the fp32 forward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pfnl_spec
from pfnl_amd import ops, synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry

def dev(a): return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
rng = np.random.default_rng(0)
for (B, T, H, W, kind) in [(1, 7, 16, 16, "u"), (2, 7, 20, 36, "u"), (1, 5, 12, 22, "u"), (1, 3, 6, 10, "u"), (1, 7, 64, 64, "u"), (1, 7, 128, 128, "u"),
                           (1, 7, 32, 32, "dark"), (1, 7, 32, 32, "const"), (1, 7, 32, 32, "peak"), (1, 7, 64, 64, "smooth")]:
    C = 12 * T
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    if kind == "dark": x *= 0.02
    if kind == "const": x[:] = 0.5
    if kind == "peak": x = (x * 0.1); x[:, :, 5, 7] = 0.97
    if kind == "smooth":
        yy, xx = np.mgrid[0:H, 0:W]
        x = (0.5 + 0.4 * np.sin(yy / 9.0)[None, None, :, :, None] * np.cos(xx / 7.0)[None, None, :, :, None] + 0.02 * x).astype(np.float32)
    wg = (rng.normal(size=(1, 1, C, C)) * 0.1).astype(np.float32); ww = (rng.normal(size=(1, 1, C, C)) * 0.1).astype(np.float32)
    bg = (rng.normal(size=C) * 0.05).astype(np.float32); bw = (rng.normal(size=C) * 0.05).astype(np.float32)
    stack = np.concatenate([x[:, t] for t in range(T)], -1).astype(np.float64)
    z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64), ww.astype(np.float64), bw.astype(np.float64), stabilise=True)
    ref = stack + pfnl_spec.depth_to_space2(z)
    errs = {p: float(np.abs(ops.nonlocal_residual(dev(x), wg, bg, ww, bw, precision=p).cpu().numpy() - ref).max()) for p in ("fp32", "split16", "bf16", "f16")}
    print("%dx%dx%3dx%3d %-6s max|err| vs fp64 (|z| max %.2f):" % (B, T, H, W, kind, np.abs(z).max()), {k: "%.2e" % v for k, v in errs.items()})

geom = PFNLGeometry()
eng = PFNLEngine(geom, device=0); eng.load_weights(synth.synthetic_weights(geom, seed=0))
for (B, H, W) in ((4, 128, 128), (1, 270, 480)):
    clips = synth.uniform_clips(B, 7, H, W, seed=8)
    xd = torch.from_numpy(clips).cuda()
    out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    eng.set_option("precision", "fp32")
    eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0); torch.cuda.synchronize()
    y32 = out.cpu().numpy().copy()
    eng.set_option("precision", "bf16")
    for nl in ("split", "f16"):
        eng.set_option("bf16_nonlocal", nl)
        for _ in range(2): eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize(); eng.profile_reset(); eng.profile(1)
        t0 = time.perf_counter()
        for _ in range(5): eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        eng.profile(0); pr = eng.profile_read(); y = out.cpu().numpy()
        mse = float(np.mean((y.astype(np.float64) - y32) ** 2))
        print("%dx7x%dx%d bf16, bf16_nonlocal=%-5s: %.3f ms/step, nl_attn %.3f ms; vs fp32 forward: max|d| %.2e, PSNR %.1f dB" % (
            B, H, W, nl, dt * 1e3, pr["nl_attn"]["ms"] / 5, np.abs(y - y32).max(), 10 * np.log10(1.0 / mse)))
