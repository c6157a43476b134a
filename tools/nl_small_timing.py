import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pfnl_amd.engine import PFNLEngine
from pfnl_amd.spec import PFNLGeometry
from pfnl_amd import synth
g = PFNLGeometry()
for nl in ("auto", "split16", "f32"):
    eng = PFNLEngine(g)
    eng.load_weights(synth.synthetic_weights(g, seed=0))
    eng.set_option("nonlocal", nl)
    x = torch.from_numpy(synth.uniform_clips(1, 7, 32, 32, seed=1)).cuda()
    out = torch.empty(eng.out_shape(1, 32, 32), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        eng.forward_device(x.data_ptr(), out.data_ptr(), 1, 32, 32, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        eng.forward_device(x.data_ptr(), out.data_ptr(), 1, 32, 32, st)
    torch.cuda.synchronize()
    print(nl, "%.4f ms" % ((time.perf_counter() - t0) / 300 * 1e3))
    eng.close()
