"""Phase timeline of conv3x3_split16_kernel (variant build -DPFNL_S16_TIMING -> pfnl_amd/lib/var_s16_T.so): 9 stamps per
unit (s_memtime cycles): 0 unit start, 1 groups 0-1 done, 2 past b0, 3 weight slice written, 4 groups 2-3 done, 5 past b1,
6 weight slice written, 7 half the halo committed, 8 groups 4-5 done.
usage: PFNL_HIP_LIB=pfnl_amd/lib/var_s16_T.so python tools/s16_timing.py [fused]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops, _capi
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
T, H, W = 28, 128, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(T, H, W, 64, generator=g).cuda()
k = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
b = np.zeros(64, np.float32)
kw = dict(addend=torch.randn(T // 7, H, W, 64, generator=g).cuda(), add_div=7, resid=torch.randn(T, H, W, 64, generator=g).cuda()) if fused else {}
for _ in range(2):
    ops.conv3x3_winograd(x, k, b, act=True, variant="split16", **kw)
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 2 * 128, np.int64)
lib.pfnl_debug_read_s16_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_s16_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 2, 128)
names = ["g0-1", "b0 wait", "w-write", "g2-3", "b1 wait", "w-write", "commit A", "g4-5(+commit B)", "b2 wait+next start"]
for wg in (0, 9, 100, 255):
    for wi, wname in ((0, "wave0"), (1, "wave5")):
        s = st[wg, wi]
        n = int((s != 0).sum()) // 9
        if not n:
            continue
        s = s[:n * 9].reshape(n, 9)
        print(f"wg {wg} {wname}: {n} units recorded; unit period: {(s[1:, 0] - s[:-1, 0]).tolist()}")
        for i in range(min(n, 6)):
            row = s[i]
            nxt = s[i + 1, 0] if i + 1 < n else row[8]
            d = [row[1] - row[0], row[2] - row[1], row[3] - row[2], row[4] - row[3], row[5] - row[4], row[6] - row[5], row[7] - row[6], row[8] - row[7], nxt - row[8]]
            print("   unit %d (%s): " % (i, "A" if i % 2 == 0 else "B") + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, d)))
