"""Same-box A/B of library builds (PFNL_HIP_LIB=pfnl_amd/lib/var_<name>.so): one JSON line per run with the sustained ms per step,
the package power and shader clock rocm-smi reported DURING the run (the split-f16 launches sit on the power cap: a variant is only
understood with W and MHz next to its microseconds), the per-class kernel times (HIP events, sampled blocks) and a sha256 of the
output (variants that only reorder instructions must be bit-identical).  Usage: python tools/ab_r05.py [cfg2|cfg4|cfg4bf16|cfg0|cfg5] [seconds]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
    import torch
    import bench
    from pfnl_amd import synth
    from pfnl_amd.engine import PFNLEngine
    from pfnl_amd.spec import PFNLGeometry
    shapes = {"cfg2": (4, 7, 128, 128, 4, False), "cfg4": (1, 7, 270, 480, 4, False), "cfg4bf16": (1, 7, 270, 480, 4, True),
              "cfg0": (1, 7, 32, 32, 4, False), "cfg5": (1, 5, 64, 64, 2, False)}
    B, T, H, W, scale, bf16 = shapes[wl]
    geom = PFNLGeometry(num_frames=T, scale=scale)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(synth.synthetic_weights(geom, seed=0))
    if bf16:
        eng.set_option("precision", "bf16")
    for kv in os.environ.get("AB_OPTIONS", "").split(","):
        if "=" in kv:
            eng.set_option(*kv.split("=", 1))
    x = torch.from_numpy(synth.uniform_clips(B, T, H, W, seed=1234)).cuda()
    out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    step = lambda: eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, st)   # noqa: E731
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:                             # clock ramp
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 20
    n = max(20, int(seconds / est))
    sampler = bench.PowerSampler(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    power = sampler.stop() or {}
    small = wl in ("cfg0", "cfg5")
    eng.profile_reset()
    eng.profile(4 if small else 2)
    m = 50 if small else 10
    for _ in range(m):
        step()
    torch.cuda.synchronize()
    eng.profile(False)
    prof = eng.profile_read()
    nb = geom.num_block
    sc = 1.0 if small else nb / float(bench.blocks_sampled(nb))
    kms = {k: round(v["ms"] / m * (sc if k in ("conv3x3", "conv1x1") else 1.0), 4) for k, v in prof.items() if v["ms"] > 0}
    print(json.dumps({"label": os.environ.get("AB_LABEL", os.path.basename(os.environ.get("PFNL_HIP_LIB", "product"))), "workload": wl,
                      "ms_per_step": round(1e3 * el / n, 4), "fps": round(B * n / el, 1), "steps": n,
                      "package_w": power.get("package_w"), "sclk_mhz": power.get("sclk_mhz"), "power_samples": power.get("samples"),
                      "joules_per_frame": round(power["package_w"] * el / n / B, 4) if power.get("package_w") else None,
                      "kernel_ms": kms, "out_sha": sha}), flush=True)


if __name__ == "__main__":
    main()
