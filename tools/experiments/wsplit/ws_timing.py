"""Phase timeline of conv_wsplit_kernel (variant build -DPFNL_WS_TIMING -> pfnl_amd/lib/var_wsT.so): per M-block 8 stamps per wave
(shader cycles): 0 block start, 1 DMA issued, 2 pipeline filled, 3 k-steps done, 4 z written, 5 vmcnt(0) passed, 6 past the barrier,
7 output stage issued.  usage: PFNL_HIP_LIB=pfnl_amd/lib/var_wsT.so python tools/ws_timing.py [fused]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import ops, _capi
fused = len(sys.argv) > 1
items, H, W = 28, 128, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(items, H, W, 64, generator=g).cuda()
k = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
b = np.zeros(64, np.float32)
kw = dict(addend=torch.randn(items // 7, H, W, 64, generator=g).cuda(), add_div=7, resid=torch.randn(items, H, W, 64, generator=g).cuda()) if fused else {}
for _ in range(3):
    ops.conv3x3_winograd(x, k, b, variant="wsplit", **kw)
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 4 * 256, np.int64)
lib.pfnl_debug_read_ws_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_ws_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 4, 256)
fine = os.environ.get("WS_FINE") == "1"
names = ["dma issue", "fill", "k-steps", "z write", "vmcnt", "barrier", "output", "closing barrier"]
if fine:
    names = ["dma issue", "fill", "steps 0-4", "step 5", "step 6", "step 7", "pass end 1", "steps 8-15", "z write", "vmcnt", "barrier", "output", "closing barrier"]
NS = len(names)
for wg in (0, 9, 100, 255):
    for wv in (0, 3):
        s = st[wg, wv]
        n = (int((s != 0).sum()) - 2) // NS
        if n <= 0:
            continue
        print(f"wg {wg} wave {wv}: {n} blocks; entry -> first block {int(s[1] - s[0])}; block period {[int(s[1 + NS * (i + 1)] - s[1 + NS * i]) for i in range(n - 1)]}")
        for i in range(min(n, 4)):
            row = s[1 + NS * i: 1 + NS * i + NS + 1]
            print(f"   block {i}: " + "  ".join(f"{nm} {int(row[j + 1] - row[j])}" for j, nm in enumerate(names)))
