"""The C-ABI library loads without a GPU and exports every symbol include/pfnl_hip.h declares;
argument validation that needs no device works; compute without a device fails loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from pfnl_amd import _capi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pfnl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfnl_[a-z0-9_]+)\s*\(", text)))


def test_library_built_and_exports_header_symbols():
    assert os.path.exists(_capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(_capi.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"libpfnl_hip.so does not export {s}"
    assert set(syms) == set(_capi.SIGNATURES), "ctypes signatures and header disagree"


def test_version_and_argument_validation_without_gpu():
    lib = _capi.load_library()
    assert lib.pfnl_version() == 3
    h = C.c_void_p()
    bad = _capi.pfnl_config(4, 4, 64, 20, 0, (C.c_int32 * 3)(0, 0, 0))        # even num_frames
    assert lib.pfnl_create(C.byref(bad), C.byref(h)) == -1
    assert b"num_frames" in lib.pfnl_last_error()
    bad = _capi.pfnl_config(7, 3, 64, 20, 0, (C.c_int32 * 3)(0, 0, 0))        # scale 3
    assert lib.pfnl_create(C.byref(bad), C.byref(h)) == -1
    assert lib.pfnl_create(None, C.byref(h)) == -1


def test_no_cpu_fallback():
    """Without a device the product path must fail, not quietly compute somewhere else."""
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    from pfnl_amd.engine import PFNLEngine
    with pytest.raises(_capi.PFNLHipError, match="no HIP device"):
        PFNLEngine()
    from model.pfnl import PFNL
    with pytest.raises(_capi.PFNLHipError):
        PFNL().forward(np.zeros((1, 7, 8, 8, 3), np.float32))


def test_no_store_data_hazard_in_emitted_isa():
    """tools/lint_store_hazard.py over every kernel source: no VALU write to the data registers of a 16-byte buffer store
    with an SGPR offset within two issue slots (pfnl_amd/csrc/common.h, buffer_store_b128_guarded); and (round 5) no inline-asm
    instruction reading a register an MFMA wrote fewer than 11 issue slots earlier, no instruction at all reading the result of an
    inline-asm MFMA inside that window - the compiler pads neither (a fold of accumulators written as inline asm read them one MFMA
    after they were issued: repeatable at -O3, run-to-run noise in the -O1 ASAN build)."""
    import shutil
    import subprocess
    import sys

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    st = subprocess.run([sys.executable, os.path.join(root, "tools", "lint_store_hazard.py"), "--selftest"], capture_output=True, text=True)
    assert st.returncode == 0, st.stdout + st.stderr                 # the rule fires on known-bad snippets (VGPR and AGPR destinations), not on good ones
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "lint_store_hazard.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
