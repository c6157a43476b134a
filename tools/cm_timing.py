"""Phase timeline of conv_small_kernel workgroups (build: tools/build_variant.sh cmt conv_small.hip -DPFNL_CM_TIMING).
Stamps (100 MHz wall clock): 0 start, 1 halo requested, 2 halo committed, 3 past the first barrier, 4 first source's MFMAs issued,
5 K loop done, 6 K halves met, 7 end.   usage: PFNL_HIP_LIB=.../var_cmt.so python tools/cm_timing.py [T clips H W]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import _capi, ops
T, clips, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (7, 1, 32, 32)
lib = _capi.load_library()
rng = np.random.default_rng(0)
F = clips * T
x = torch.from_numpy(rng.normal(size=(F, H, W, 64)).astype(np.float32)).cuda()
base = torch.from_numpy(rng.normal(size=(clips, H, W, 64)).astype(np.float32)).cuda()
res = torch.from_numpy(rng.normal(size=(F, H, W, 64)).astype(np.float32)).cuda()
k1 = (rng.normal(size=(3, 3, 64, 64)) / 24).astype(np.float32)
k2 = (rng.normal(size=(3, 3, 128, 64)) / 34).astype(np.float32)
b = np.zeros(64, np.float32)
fn = lib.pfnl_debug_read_cm_stamps
for name, call in (("conv1", lambda: ops.conv_small(x, k1, b)), ("conv2", lambda: ops.conv_small(x, k2, b, a=base, a_div=T, resid=res))):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    n = 4096 * 16
    buf = (C.c_longlong * n)()
    fn(buf, C.c_size_t(n))
    a = np.frombuffer(buf, dtype=np.int64).reshape(4096, 16)
    nb = int((a[:, 7] > 0).sum())
    a = a[:nb]
    t0 = a[:, 0].min()
    d = (a[:, :8] - a[:, :1]) / 100.0          # us since the workgroup's own start
    print("%s T%d %dx%dx%d: %d workgroups; start spread %.2f us; last end %.2f us after first start" % (name, T, clips, H, W, nb, (a[:, 0].max() - t0) / 100.0, (a[:, 7].max() - t0) / 100.0))
    print("   median us since own start: " + "  ".join("%d:%.2f" % (i, np.median(d[:, i])) for i in range(8)))
    print("   p90    us since own start: " + "  ".join("%d:%.2f" % (i, np.percentile(d[:, i], 90)) for i in range(8)))
