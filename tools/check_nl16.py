"""Non-local block: f32 MFMA kernel vs split-f16 kernel vs bf16-split kernel against the fp64 spec, and timing at configs[1] / 1080p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pfnl_spec
from pfnl_amd import ops, synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry

def dev(a): return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
rng = np.random.default_rng(0)
for (B, T, H, W, kind) in [(1, 7, 16, 16, "u"), (2, 7, 20, 36, "u"), (1, 5, 12, 22, "u"), (1, 3, 6, 10, "u"), (1, 7, 64, 64, "u"), (1, 7, 32, 32, "dark"), (1, 7, 32, 32, "const"), (1, 7, 32, 32, "peak")]:
    C = 12 * T
    x = rng.random((B, T, H, W, 3), dtype=np.float32)
    if kind == "dark": x *= 0.02
    if kind == "const": x[:] = 0.5
    if kind == "peak": x = (x * 0.1); x[:, :, 5, 7] = 0.97
    wg = (rng.normal(size=(1, 1, C, C)) * 0.1).astype(np.float32); ww = (rng.normal(size=(1, 1, C, C)) * 0.1).astype(np.float32)
    bg = (rng.normal(size=C) * 0.05).astype(np.float32); bw = (rng.normal(size=C) * 0.05).astype(np.float32)
    stack = np.concatenate([x[:, t] for t in range(T)], -1).astype(np.float64)
    z = pfnl_spec.nonlocal_block(pfnl_spec.space_to_depth2(stack), wg.astype(np.float64), bg.astype(np.float64), ww.astype(np.float64), bw.astype(np.float64), stabilise=True)
    ref = stack + pfnl_spec.depth_to_space2(z)
    errs = {p: float(np.abs(ops.nonlocal_residual(dev(x), wg, bg, ww, bw, precision=p).cpu().numpy() - ref).max()) for p in ("fp32", "split16", "bf16")}
    print("%dx%dx%dx%d %-5s max|err| vs fp64:" % (B, T, H, W, kind), {k: "%.2e" % v for k, v in errs.items()})
geom = PFNLGeometry()
eng = PFNLEngine(geom, device=0); eng.load_weights(synth.synthetic_weights(geom, seed=0))
for (B, H, W) in ((4, 128, 128), (1, 270, 480)):
    xd = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=8)).cuda()
    out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    ys = {}
    for nl in ("f32", "split16"):
        eng.set_option("nonlocal", nl)
        for _ in range(2): eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize(); eng.profile_reset(); eng.profile(1)
        t0 = time.perf_counter()
        for _ in range(5): eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        eng.profile(0); pr = eng.profile_read(); ys[nl] = out.cpu().numpy().copy()
        print("%dx7x%dx%d nonlocal=%-7s: %.3f ms/step, nl_attn %.3f ms, nl_pack %.3f" % (B, H, W, nl, dt * 1e3, pr["nl_attn"]["ms"] / 5, pr["nl_pack"]["ms"] / 5))
    print("   max|forward(split16) - forward(f32)| = %.3g" % np.abs(ys["split16"] - ys["f32"]).max())
