"""Activated by fixtures written by tools/make_tf1_golden.py (the REFERENCE's graph under TensorFlow 1.x, see that script).

Absent fixtures = skipped: the build image has no TensorFlow, so today parity is pinned only on the oracle
(oracle/README.md, "parity unpinned").  With `tests/golden/tf1_<case>.npz` present these tests pin
  * the oracle (fp32 restatement) against the reference's outputs,
  * the variable names / shapes of pfnl_amd/spec.py against what TensorFlow created,
  * the HIP path against the reference's outputs (|dPSNR| <= 0.01 dB is implied by the element-wise bound)."""
import ast
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(f for f in glob.glob(os.path.join(ROOT, "tests", "golden", "tf1_*.npz")) if "checkpoint_index" not in f)
needs_fixture = pytest.mark.skipif(not FIXTURES, reason="no tests/golden/tf1_*.npz (run tools/make_tf1_golden.py where TF 1.12 exists)")
TOL = 5e-5          # element-wise on [0,1]-scale outputs (tests/test_gpu_forward.py ABS_TOL)


def _load(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@needs_fixture
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f) for f in FIXTURES])
def test_variable_names_match_tensorflow(path):
    from pfnl_amd.spec import PFNLGeometry
    d = _load(path)
    tf_names = {str(n)[:-2] if str(n).endswith(":0") else str(n): ast.literal_eval(str(s)) for n, s in zip(d["var_names"], d["var_shapes"])}
    want = dict(PFNLGeometry().weight_shapes())
    assert set(tf_names) == set(want), (sorted(set(tf_names) ^ set(want)))
    for k, shp in want.items():
        assert tuple(tf_names[k]) == tuple(shp), k


@needs_fixture
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f) for f in FIXTURES])
def test_oracle_matches_tensorflow(path):
    from oracle import pfnl_fast
    from pfnl_amd import synth
    from pfnl_amd.spec import PFNLGeometry
    d = _load(path)
    w = synth.synthetic_weights(PFNLGeometry(), seed=int(d["weight_seed"]))
    y = pfnl_fast.FastOracle(w, num_block=20).forward(d["x"])
    err = float(np.abs(y - d["y"]).max())
    print(os.path.basename(path), "max|oracle - TF1| =", err)
    assert y.shape == d["y"].shape and err < TOL, err


@needs_fixture
@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f) for f in FIXTURES])
def test_hip_matches_tensorflow(path):
    from pfnl_amd import synth
    from pfnl_amd.engine import PFNLEngine
    from pfnl_amd.spec import PFNLGeometry
    d = _load(path)
    g = PFNLGeometry()
    eng = PFNLEngine(g, device=0)
    eng.load_weights(synth.synthetic_weights(g, seed=int(d["weight_seed"])))
    y = eng.forward(d["x"])
    eng.close()
    err = float(np.abs(y - d["y"]).max())
    print(os.path.basename(path), "max|hip - TF1| =", err, "PSNR", synth.psnr(y, d["y"]))
    assert err < TOL, err


def test_checkpoint_index_names_if_present():
    """Names found in the published checkpoint's .index (recorded by make_tf1_golden.py --checkpoint) cover the spec's."""
    path = os.path.join(ROOT, "tests", "golden", "tf1_checkpoint_index.npz")
    if not os.path.exists(path):
        pytest.skip("no tf1_checkpoint_index.npz")
    from pfnl_amd.spec import PFNLGeometry
    d = _load(path)
    have = {str(n): ast.literal_eval(str(s)) for n, s in zip(d["names"], d["shapes"])}
    for k, shp in PFNLGeometry().weight_shapes():
        assert k in have and tuple(have[k]) == tuple(shp), k
