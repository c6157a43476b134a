"""A few forwards of the default engine (for rocprofv3 runs: tools/sf_variants.sh).  usage: run_fwd.py B H W [precision]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pfnl_amd import synth  # noqa: E402
from pfnl_amd.engine import PFNLEngine  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 128, 128)
geom = PFNLGeometry()
eng = PFNLEngine(geom, device=0)
eng.load_weights(synth.synthetic_weights(geom, seed=0))
if len(sys.argv) > 4:
    eng.set_option("precision", sys.argv[4])
xd = torch.from_numpy(synth.uniform_clips(B, 7, H, W, seed=8)).cuda()
out = torch.empty(eng.out_shape(B, H, W), dtype=torch.float32, device="cuda")
for _ in range(8):
    eng.forward_device(xd.data_ptr(), out.data_ptr(), B, H, W, 0)
torch.cuda.synchronize()
