"""Phase timeline of conv3x3_bf16_kernel (variant build -DPFNL_BF16_TIMING -> pfnl_amd/lib/var_bt.so): 16 stamps per tile -
the start of each of the 12 groups, the commit pair inside group 8 and the pair around the closing barrier.
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
NS = 256
buf = np.zeros(256 * 8 * NS, np.int64)
lib.pfnl_debug_read_bf16_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_bf16_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 8, NS)
# stamp order within a tile: g0 g1 g2 g3 g4 g5 g6 g7 g8 [commit start, commit end] g9 g10 g11 [before barrier, after barrier]
names = ["g0", "g1", "g2", "g3", "g4(b)", "g5", "g6", "g7", "g8(b)..commit", "commit", "rest g8", "g9", "g10(req)", "g11", "barrier wait", "->next g0"]
for wg in (0, 100):
    for wave in (0, 4, 5):
        s = st[wg, wave]
        n = int((s != 0).sum()) // 16
        s = s[:n * 16].reshape(n, 16)
        per = s[1:, 0] - s[:-1, 0]
        print(f"wg {wg} wave {wave}: {n} tiles; tile period: {per.tolist()}")
        d = np.diff(s.reshape(-1)).reshape(-1)
        d = np.concatenate([d, [0]]).reshape(n, 16)
        for i in (2, 5, 8):
            if i < n:
                print("   tile %d: " % i + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, d[i])))
        if n > 3:
            print("   mean over tiles 1..%d: " % (n - 2) + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, d[1:n - 1].mean(axis=0))))
