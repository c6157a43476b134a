"""Forward time against the launch structure for shapes between the small-shape trunk (< 256 tiles of 8x32 pixels per 3x3 launch) and
the headline batch: which of the existing structures should `auto` take where?  fp32: 2 launches per block (chains of T frames per
workgroup), 3 (split16_c10=off), 4 with chains (+ split16_chain=off), 4 dealt out tile by tile (what split16_mid=auto takes below 136 chains), the
small-shape kernels (small=on); bf16 for comparison.
usage: python tools/precision_ladder.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth

SHAPES = ((1, 32, 32), (1, 96, 96), (1, 112, 112), (1, 128, 128), (1, 144, 180), (1, 160, 224), (1, 192, 192), (1, 180, 318), (1, 270, 480), (2, 96, 96),
          (2, 128, 128), (2, 144, 180), (3, 128, 128), (4, 64, 64), (4, 128, 128), (8, 64, 64))
VARIANTS = (("auto", {}), ("2/blk", {"split16_mid": "off"}), ("3/blk", {"split16_mid": "off", "split16_c10": "off"}),
            ("4/blk chains", {"split16_mid": "off", "split16_c10": "off", "split16_chain": "off"}), ("4/blk flat", {"env": {"PFNL_SF_MID_CHAINS": "1000000"}}),
            ("small", {"small": "on"}), ("bf16", {"precision": "bf16"}), ("bf16 chains", {"precision": "bf16", "split16_mid": "off"}))
geom = PFNLGeometry()
res = {}
for name, opts in VARIANTS:
    for k, v in opts.get("env", {}).items():
        os.environ[k] = v
    eng = PFNLEngine(geom)
    for k in opts.get("env", {}):
        del os.environ[k]
    eng.load_weights(synth.synthetic_weights(geom, seed=0))
    for k, v in opts.items():
        if k != "env":
            eng.set_option(k, v)
    for B, H, W in SHAPES:
        if name == "small" and B * H * W > 4 * 128 * 128:
            continue
        x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
        out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
        for _ in range(3):
            eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            eng.forward_device(x.data_ptr(), out.data_ptr(), B, H, W, 0)
        torch.cuda.synchronize()
        res[(name, B, H, W)] = 1e3 * (time.perf_counter() - t0) / n
    eng.close()
for B, H, W in SHAPES:
    tiles = B * 7 * ((H + 7) // 8) * ((W + 31) // 32)
    print("B=%d %3dx%3d tiles8x32=%5d chains=%4d  " % (B, H, W, tiles, tiles // 7)
          + "  ".join("%s %7.3f" % (n, res[(n, B, H, W)]) if (n, B, H, W) in res else "%s       -" % n for n, _ in VARIANTS), flush=True)
