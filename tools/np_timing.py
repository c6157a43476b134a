"""Phase timeline of nl_attn_f16_sw_kernel (variant build -DPFNL_NP_TIMING=0|1 (hi only | split) -> pfnl_amd/lib/var_np_T.so): per key tile
4 stamps (shader cycles) of waves 0 and 4 (the two waves of SIMD 0): tile starts, half-body a issued, the barrier between the half-bodies
passed, half-body b issued.
usage: PFNL_HIP_LIB=pfnl_amd/lib/var_np_T.so python tools/np_timing.py [bf16|f32]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import _capi, synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
geom = PFNLGeometry(num_block=1)
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    eng.set_option("precision", "bf16")
x = torch.from_numpy(synth.uniform_clips(1, 7, 270, 480, seed=1)).cuda()
for _ in range(2):
    eng.forward(x)
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 2 * 128, np.int64)
lib.pfnl_debug_read_np_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_np_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(256, 2, 128)
names = ["a: softmax(2t) | S^T(2t+1) | P V(2t-1)", "wait + barrier", "b: softmax(2t+1) | S^T(2t+2) | P V(2t) + requests", "loop"]
for wg in (0, 9, 100):
    for wi, wname in ((0, "wave0"), (1, "wave4")):
        s = st[wg, wi]
        n = int((s != 0).sum()) // 4
        if not n:
            continue
        s = s[:n * 4].reshape(n, 4)
        print(f"wg {wg} {wname}: {n} tiles stamped; tile period {(s[1:, 0] - s[:-1, 0]).tolist()[:12]}")
        for i in range(2, min(n - 1, 7)):
            row = s[i]
            d = [row[j + 1] - row[j] for j in range(3)] + [s[i + 1, 0] - row[3]]
            print("   tile %d: " % i + "  ".join(f"{nm} {int(v)}" for nm, v in zip(names, d)))
