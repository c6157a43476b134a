"""Phase timeline of conv3x3_bf16_v2_kernel (variant build -DPFNL_B2_TIMING -> pfnl_amd/lib/var_b2_T.so): per tile 8 stamps (shader
cycles): 0 tile start, 1 requests issued, 2 MFMA groups issued, 3 fence load passed (next halo landed), 4 barrier, 5 epilogue pieces
written, 6 barrier, 7 stores issued; the next tile's 0 follows the closing barrier.
usage: PFNL_HIP_LIB=pfnl_amd/lib/var_b2_T.so python tools/b2_timing.py [mode]      mode 0 plain, 1 fused, 2 conv1_i + conv10_i (default)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import _capi, synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
geom = PFNLGeometry(num_block=1)
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
eng.set_option("precision", "bf16")
x = torch.from_numpy(synth.uniform_clips(1, 7, 270, 480, seed=1)).cuda()
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 2 * 128, np.int64)
GEN3 = os.environ.get("PFNL_BF16_V3", "0") != "0"                   # conv_bf16_v3.hip (-DPFNL_B3_TIMING): 8 stamps per tile, other phases
reader = lib.pfnl_debug_read_b3_stamps if GEN3 else lib.pfnl_debug_read_b2_stamps
reader.argtypes = [C.c_void_p, C.c_size_t]
assert reader(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 2, 128)
names = (["groups 0-5", "barrier", "groups 6-11", "barrier", "requests+dump", "barrier", "lines+conv10+inputs", "closing barrier"] if GEN3 else
         ["requests", "MFMA groups", "fence wait", "barrier", "epilogue", "barrier", "lines+stores", "closing barrier"])
# (the stamps of the LAST launch that wrote them: the per-frame half of conv2_i - mode 1 - in a one-block forward)
for wg in (0, 9, 100, 255):
    for wi, wname in ((0, "wave0"), (1, "wave5")):
        s = st[wg, wi]
        entry, last = int(s[0]), int(s[s != 0][-1])
        npro = int(os.environ.get("PFNL_B3_PSTAMPS", "0"))              # (-DPFNL_B3_PSTAMPS: 5 more stamps inside the prologue)
        if npro:
            print(f"wg {wg} {wname} prologue: " + " ".join(str(int(s[i + 1] - s[i])) for i in range(npro)))
        s = s[1 + npro:]
        n = int((s != 0).sum()) // 8
        if not n:
            continue
        s = s[:n * 8].reshape(n, 8)
        print(f"wg {wg} {wname}: {n} tiles; entry -> first tile {int(s[0, 0]) - entry}, entry -> last stamp {last - entry}; first stamp {int(s[0, 0] - st[wg, 0, 1])}; tile period {(s[1:, 0] - s[:-1, 0]).tolist()}")
        for i in range(min(n, 6)):
            row = s[i]
            nxt = s[i + 1, 0] if i + 1 < n else row[7]
            d = [row[j + 1] - row[j] for j in range(7)] + [nxt - row[7]]
            print("   tile %d: " % i + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, d)))
