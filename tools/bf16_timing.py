"""Phase timeline of conv3x3_bf16_kernel (variant build -DPFNL_BF16_TIMING -> pfnl_amd/lib/var_bt.so):
per tile 6 stamps: burst start, after the LDS commit, after the epilogue, after the requests, before / after the barrier.
usage: PFNL_HIP_LIB=pfnl_amd/lib/var_bt.so python tools/bf16_timing.py [fused]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops, _capi
fused = len(sys.argv) > 1 and sys.argv[1] == "fused"
T, H, W = 7, 270, 480
g = torch.Generator().manual_seed(0)
x = torch.randn(T, H, W, 64, generator=g).to(torch.bfloat16).cuda()
k = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
b = np.zeros(64, np.float32)
kw = dict(addend=torch.randn(1, H, W, 64, generator=g).to(torch.bfloat16).cuda(), add_div=T,
          resid=torch.randn(T, H, W, 64, generator=g).to(torch.bfloat16).cuda()) if fused else {}
for _ in range(2):
    ops.conv3x3_bf16(x, k, b, act=True, **kw)
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 8 * 64, np.int64)
lib.pfnl_debug_read_bf16_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_bf16_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 8, 64)
names = ["compute->burst", "commit", "epilogue", "requests", "burst->barrier", "barrier wait"]
for wg in (0, 1, 100):
    for wave in (0, 4):
        s = st[wg, wave]
        n = int((s != 0).sum()) // 6
        s = s[:n * 6].reshape(n, 6)
        d = np.diff(np.concatenate([[s[0, 0]], s.reshape(-1)]))[1:] if n else []
        per = s[1:, 0] - s[:-1, 0]
        print(f"wg {wg} wave {wave}: {n} tiles; tile period (cycles of the 100 MHz counter? raw): {per.tolist()}")
        for i in range(min(n, 6)):
            row = s[i]
            prev_end = s[i - 1, 5] if i else row[0]
            print("   tile %d: " % i + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, [row[0] - prev_end, row[1] - row[0], row[2] - row[1], row[3] - row[2], row[4] - row[3], row[5] - row[4]])))
