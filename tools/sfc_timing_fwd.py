"""Phase timeline of the chain launch of conv2_i inside a one-block forward (variant build -DPFNL_SFC_TIMING -> pfnl_amd/lib/var_sfcT.so), for the 32x32x16
kernel (split16_mfma=32) and the 16x16x32 one (=16): per unit 8 stamps (shader cycles) of waves 0 and 5 - 0 unit start, 1 column tap 0 done, 2 fence passed,
3 past b0, 4 column tap 1 done, 5 past b1, 6 column tap 2 done, 7 closing fences passed (then b2).
usage: PFNL_HIP_LIB=pfnl_amd/lib/var_sfcT.so python tools/sfc_timing_fwd.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pfnl_amd import _capi, synth
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
geom = PFNLGeometry(num_block=1)
x = torch.from_numpy(synth.uniform_clips(4, 7, 128, 128, seed=1)).cuda()
lib = _capi.load_library()
lib.pfnl_debug_read_sfc_stamps.argtypes = [C.c_void_p, C.c_size_t]
names = ["tap 0", "fence wait", "b0", "tap 1", "b1", "tap 2", "closing fences", "b2 -> next unit"]
for mfma in ("32", "16"):
    eng = PFNLEngine(geom)
    eng.load_weights(synth.synthetic_weights(geom, seed=0))
    eng.set_option("split16_mfma", mfma)
    for _ in range(3):
        eng.forward(x)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 2 * 160, np.int64)
    assert lib.pfnl_debug_read_sfc_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    st = buf.reshape(256, 2, 160)
    tot = []
    for wg in range(0, 256, 17):
        s = st[wg, 0]
        n = int((s != 0).sum()) // 8
        if n < 3:
            continue
        s = s[:n * 8].reshape(n, 8)
        d = np.concatenate([s[:-1, 1:] - s[:-1, :-1], (s[1:, 0] - s[:-1, 7])[:, None]], axis=1)
        tot.append(d[1:].mean(axis=0))
    m = np.mean(tot, axis=0)
    print("split16_mfma=%s: unit period %.0f cycles;  " % (mfma, m.sum()) + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, m)))
    eng.close()
