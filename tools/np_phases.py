"""Phases of a workgroup of nl_attn_f16_sw_kernel (variant build -DPFNL_NP_TIMING=0|1 (hi only | split) -> pfnl_amd/lib/var_np_T.so): entry -> the loop
starts (query operands, the first tiles of the ring) -> the loop is through -> the last P V, ring drained -> W' staged in LDS -> projection and stores issued.
Constant-clock ticks (10 ns), medians over the workgroups of key split 0 of clip 0.
usage: PFNL_HIP_LIB=pfnl_amd/lib/var_np_T.so python tools/np_phases.py [B] [H] [W] [bf16|fp32]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import _capi, synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W = int(sys.argv[3]) if len(sys.argv) > 3 else 128
geom = PFNLGeometry(num_block=1)
eng = PFNLEngine(geom)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
if len(sys.argv) > 4 and sys.argv[4] == "bf16":
    eng.set_option("precision", "bf16")
x = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=1)).cuda()
for _ in range(3):
    eng.forward(x)
torch.cuda.synchronize()
lib = _capi.load_library()
buf = np.zeros(256 * 8, np.int64)
lib.pfnl_debug_read_np_phases.argtypes = [C.c_void_p, C.c_size_t]
assert lib.pfnl_debug_read_np_phases(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
ph = buf.reshape(256, 8)
ph = ph[ph[:, 5] != 0][:, :6]
d = np.diff(ph, axis=1) * 0.01                                     # us
names = ["prologue (queries, first tiles)", "key-tile loop", "last P V + ring drained", "W' -> LDS, O^T split", "projection + stores issued"]
print("%dx7x%dx%d: %d workgroups stamped; plan nl=%s" % (B, H, W, len(ph), eng.plan(B, H, W).get("nl")))
for i, nme in enumerate(names):
    print("  %-34s median %6.2f us   (min %6.2f, max %6.2f)" % (nme, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
print("  %-34s median %6.2f us" % ("entry -> last instruction", np.median((ph[:, 5] - ph[:, 0]) * 0.01)))
