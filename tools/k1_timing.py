"""Phase timeline of conv3x3_c1c10_kernel (variant build -DPFNL_S16_TIMING -> pfnl_amd/lib/var_s16_T.so): per tile 10 stamps
(shader cycles): 0 unit A start, 1 unit B start, 2 sub-steps done, 3 past b2, 4 row 0 prepared + dumped, 5 past the barrier,
6 row 0's conv10 products + stores issued and row 1 prepared, 7 past the barrier, 8 row 1 dumped + barrier, 9 row 1 done.
usage: PFNL_HIP_LIB=pfnl_amd/lib/var_s16_T.so python tools/k1_timing.py [sf0] [summary]     sf0: conv3x3_c1c10_kernel<true> (halo by LDS-DMA
from the split-format copy of inp0, round 6); summary: medians over every workgroup and tile instead of four workgroups' timelines"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops, _capi
T, clips, H, W = 7, 4, 128, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(T * clips, H, W, 64, generator=g).cuda()
k1 = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
k10 = (torch.randn(1, 1, 64 * T, 64, generator=g) * 0.05).numpy()
b = np.zeros(64, np.float32)
SF0 = "sf0" in sys.argv[1:]
for _ in range(2):
    ops.conv1_conv10_split16(x, k1, b, k10, b, T, sf0=SF0)
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 2 * 128, np.int64)
lib.pfnl_debug_read_s16_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_s16_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 2, 128)
if "summary" in sys.argv[1:]:
    rows = []
    for wg in range(256):
        for wi in (0, 1):
            r = st[wg, wi]
            n = int((r != 0).sum()) // 10
            if n >= 3:
                r = r[:n * 10].reshape(n, 10)
                for i in range(1, n - 1):                           # steady state: not the first, not the last tile
                    rows.append([r[i, j + 1] - r[i, j] for j in range(9)] + [r[i + 1, 0] - r[i, 9], r[i + 1, 0] - r[i, 0]])
    a = np.array(rows, dtype=np.float64)
    nm = ["unit A", "unit B sub-steps", "b2 wait", "prep0+dump0", "bar", "mm0+prep1+st0", "bar", "dump1+bar", "mm1+st1", "to next tile", "TILE PERIOD"]
    print("conv3x3_c1c10_kernel<%s>: %d steady-state tiles, median cycles per phase (waves 0 and 5 of every workgroup)" % ("true" if SF0 else "false", len(a)))
    print("   " + "   ".join("%s %d" % (n_, int(np.median(a[:, j]))) for j, n_ in enumerate(nm)))
    sys.exit(0)
names = ["unit A", "unit B sub-steps", "b2 wait", "prep0+dump0", "bar", "mm0+prep1+st0", "bar", "dump1+bar", "mm1+st1", "to next tile"]
for wg in (0, 9, 100, 255):
    for wi, wname in ((0, "wave0"), (1, "wave5")):
        s = st[wg, wi]
        n = int((s != 0).sum()) // 10
        if not n:
            continue
        s = s[:n * 10].reshape(n, 10)
        print(f"wg {wg} {wname}: {n} tiles; tile period {(s[1:, 0] - s[:-1, 0]).tolist()}")
        for i in range(n):
            row = s[i]
            nxt = s[i + 1, 0] if i + 1 < n else row[9]
            d = [row[j + 1] - row[j] for j in range(9)] + [nxt - row[9]]
            print(f"   tile {i}: " + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, d)) + f"   | serial {int(row[9] - row[2])}")
