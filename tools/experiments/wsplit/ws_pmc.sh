#!/bin/bash
# PMC passes of the wsplit 3x3 op (28 x 128 x 128): LDS conflicts / waits, instruction mix.  usage (GPU box): tools/ws_pmc.sh OUTDIR
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; OUT=${1:-gpurun_out/wspmc}; mkdir -p $OUT
cat > /tmp/ws_drv.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
from pfnl_amd import ops
g = torch.Generator().manual_seed(0)
x = torch.randn(28, 128, 128, 64, generator=g).cuda()
k = (torch.randn(3, 3, 64, 64, generator=g) * 0.05).numpy()
b = np.zeros(64, np.float32)
for _ in range(3):
    ops.conv3x3_winograd(x, k, b, variant="wsplit")
torch.cuda.synchronize()
PY
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --kernel-trace -d $OUT/p$i -o p -- python /tmp/ws_drv.py > $OUT/p$i.log 2>&1
  python tools/rocprof_pmc.py $(find $OUT/p$i -name "*.db" | head -1) 2>&1 | grep -E "kernel|wsplit" | tee -a $OUT/pmc.md
  rm -rf $OUT/p$i
done
