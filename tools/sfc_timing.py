"""Phase timeline of conv3x3_sf_chain_kernel (variant build -DPFNL_SFC_TIMING -> pfnl_amd/lib/var_sfcT.so): per unit 8 stamps (shader
cycles) of waves 0 and 5: 0 unit start, 1 groups 0-1 done, 2 fence (weight slice 2 + the next halo) passed, 3 past b0, 4 groups 2-3 done,
5 past b1, 6 groups 4-5 done, 7 closing fences passed (then b2).  usage: PFNL_HIP_LIB=pfnl_amd/lib/var_sfcT.so python tools/sfc_timing.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops, _capi
T, clips, H, W = 7, 4, 128, 128
g = torch.Generator().manual_seed(0)
F = T * clips
x = torch.randn(F, H, W, 64, generator=g).cuda()
base = torch.randn(clips, H, W, 64, generator=g).cuda()
res = torch.randn(F, H, W, 64, generator=g).cuda()
k = (torch.randn(3, 3, 128, 64, generator=g) * 0.05).numpy()
b = np.zeros(64, np.float32)
for _ in range(3):
    ops.conv3x3_winograd(x, k, b, addend=base, add_div=T, resid=res, variant="split16_sf_chain")
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 2 * 160, np.int64)
lib.pfnl_debug_read_sfc_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_sfc_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 2, 160)
names = ["groups 0-1", "fence wait", "b0", "groups 2-3", "b1", "groups 4-5", "closing fences", "b2 -> next unit"]
for wg in (0, 9, 100, 255):
    for wi, wname in ((0, "wave0"), (1, "wave5")):
        s = st[wg, wi]
        n = int((s != 0).sum()) // 8
        if n < 3:
            continue
        s = s[:n * 8].reshape(n, 8)
        per = s[1:, 0] - s[:-1, 0]
        print(f"wg {wg} {wname}: {n} units; unit period mean {per.mean():.0f} (min {per.min()}, max {per.max()})")
        d = np.concatenate([s[:-1, 1:] - s[:-1, :-1], (s[1:, 0] - s[:-1, 7])[:, None]], axis=1)
        print("   mean per phase: " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d.mean(axis=0))))
        print("   max  per phase: " + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, d.max(axis=0))))
