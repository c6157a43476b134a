"""Host-side logic (no GPU): drop-in class surface, harness helpers, checkpoint layout."""
import inspect
import os

import numpy as np
import pytest

from conftest import load_golden
from pfnl_amd import checkpoint, synth, tfbundle
from pfnl_amd import model as M
from pfnl_amd.spec import PFNLGeometry, check_weights


def test_class_surface_matches_reference():
    from model.pfnl import PFNL          # the import the reference's main.py:8 performs
    from model.base_model import VSR
    m = PFNL()
    assert isinstance(m, VSR)
    # reference model/pfnl.py:21-37
    want = dict(num_frames=7, scale=4, in_size=32, gt_size=128, eval_in_size=[128, 240], batch_size=16,
                eval_basz=4, learning_rate=1e-3, end_lr=1e-4, reload=True, max_step=int(1.5e5 + 1),
                decay_step=1.2e5, train_dir='./data/filelist_train.txt', eval_dir='./data/filelist_val.txt',
                save_dir='./checkpoint/pfnl', log_dir='./pfnl.txt')
    for k, v in want.items():
        assert getattr(m, k) == v, k
    sig = lambda f: list(inspect.signature(f).parameters.items())
    assert [(k, p.default) for k, p in sig(m.test_video_truth)] == [('path', inspect._empty), ('name', 'result'),
                                                                    ('reuse', False), ('part', 50)]
    assert [(k, p.default) for k, p in sig(m.test_video_lr)] == [('path', inspect._empty), ('name', 'result'),
                                                                 ('reuse', False), ('part', 50)]
    assert [(k, p.default) for k, p in sig(m.testvideos)][1:] == [('start', 0), ('name', 'pfnl')]
    assert [k for k, _ in sig(m.load)] == ['sess', 'checkpoint_dir', 'step']
    assert [k for k, _ in sig(m.save)] == ['sess', 'checkpoint_dir', 'step']
    for fn in (m.train, m.eval, m.build):
        with pytest.raises(NotImplementedError):
            fn()
    assert PFNL.testvideo is PFNL.test_video_lr


def test_windows_and_quantise_match_golden():
    gd = load_golden("harness_5x16x24_nb1")
    lrs = (gd["lr_u8"] / 255.).astype(np.float32)
    win = M.sliding_windows(lrs, 7)
    assert win.shape == (5, 7, 16, 24, 3)
    for i in range(5):
        for k in range(7):
            assert np.array_equal(win[i, k], lrs[min(max(i + k - 3, 0), 4)])
    assert np.array_equal(M.quantise(gd["sr"][:, 0]), gd["sr_u8"])
    assert M.sliding_windows(np.zeros((0, 4, 4, 3), np.float32), 7).shape == (0, 7, 4, 4, 3)
    assert M.sliding_windows(lrs[:1], 7).shape == (1, 7, 16, 24, 3)


def test_blur_decimate_matches_scipy_definition():
    import scipy.ndimage
    delta = np.zeros((13, 13))
    delta[6, 6] = 1
    blur = scipy.ndimage.gaussian_filter(delta, 1.6)          # reference utils.py:95-105
    k = synth.gaussian_kernel_1d()
    assert np.allclose(blur, np.outer(k, k), atol=1e-15)
    rng = np.random.default_rng(0)
    hr = rng.random((2, 24, 32, 3))
    lr = synth.blur_decimate(hr, 4)
    assert lr.shape == (2, 6, 8, 3)
    hp = np.pad(hr, ((0, 0), (6, 6), (6, 6), (0, 0)), mode="reflect")
    ref = np.einsum("ij,bijc->bc", blur, hp[:, 4:17, 8:21])   # output (1,2): rows 4.., cols 8..
    assert np.allclose(lr[:, 1, 2], ref, atol=1e-6)


def test_load_returns_false_without_checkpoint(tmp_path, capsys):
    m = M.PFNL()
    assert m.load(None, str(tmp_path / "nope")) is False
    assert m.load(None, str(tmp_path)) is False
    assert "ERROR" in capsys.readouterr().out


def test_checkpoint_roundtrip_tf_layout(tmp_path):
    g = PFNLGeometry(num_block=1)
    w = synth.synthetic_weights(g)
    m = M.PFNL()
    m.num_block = 1
    m.set_weights(w)
    m.save(None, str(tmp_path), 500)
    files = set(os.listdir(tmp_path))
    assert {"checkpoint", "VSR-500.index", "VSR-500.data-00000-of-00001", "VSR-500.npz"} <= files
    assert checkpoint.read_state_file(str(tmp_path)) == "VSR-500"
    os.remove(tmp_path / "VSR-500.npz")                        # force the TF bundle reader
    m2 = M.PFNL()
    m2.num_block = 1
    assert m2.load(None, str(tmp_path)) is True
    check_weights(g, m2._weights)
    for k in w:
        assert np.array_equal(m2._weights[k], w[k]), k
    # optimizer slots / global_step in a training checkpoint are ignored; outer scope may differ
    extra = {("model/" + k): v for k, v in w.items()}
    extra["beta1_power"] = np.float32(0.9)
    extra["Variable"] = np.int64(1500)
    extra["model/nlvsr/conv0/kernel/Adam"] = np.zeros_like(w["nlvsr/conv0/kernel"])
    tfbundle.write_bundle(str(tmp_path / "VSR-1500"), {k: np.asarray(v) for k, v in extra.items()})
    got = checkpoint.load_checkpoint(str(tmp_path), g, step=1500)
    assert got is not None and np.array_equal(got[1]["nlvsr/conv0/kernel"], w["nlvsr/conv0/kernel"])


def test_tfbundle_known_answers(tmp_path):
    assert tfbundle.crc32c(b"123456789") == 0xE3069283        # CRC-32C check value
    assert tfbundle.crc32c(b"\x00" * 32) == 0x8A9136AA        # RFC 3720 B.4
    assert tfbundle.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfbundle.unmask_crc(tfbundle.mask_crc(0x12345678)) == 0x12345678
    t = {"a/b": np.arange(6, dtype=np.float32).reshape(2, 3), "a/c": np.array(7, np.int64),
         "z": np.zeros((0, 4), np.float32)}
    tfbundle.write_bundle(str(tmp_path / "x"), t)
    idx = tfbundle.read_index(str(tmp_path / "x"))
    assert idx["a/b"]["shape"] == (2, 3) and idx["a/b"]["dtype"] == tfbundle.DT_FLOAT
    back = tfbundle.read_bundle(str(tmp_path / "x"), verify_data=True)
    for k in t:
        assert np.array_equal(back[k], t[k]) and back[k].shape == t[k].shape
    raw = bytearray(open(tmp_path / "x.index", "rb").read())
    raw[5] ^= 0xFF
    open(tmp_path / "x.index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        tfbundle.read_index(str(tmp_path / "x"))


def test_many_tensor_bundle_spans_several_blocks(tmp_path):
    rng = np.random.default_rng(1)
    t = {f"scope/layer_{i:03d}/kernel": rng.random((3, 3, 2, 2)).astype(np.float32) for i in range(300)}
    tfbundle.write_bundle(str(tmp_path / "big"), t, block_size=512)
    back = tfbundle.read_bundle(str(tmp_path / "big"))
    assert set(back) == set(t) and all(np.array_equal(back[k], t[k]) for k in t)


def test_metrics_known_answers():
    from pfnl_amd import metrics as mt
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(24, 32, 3)).astype(np.float64)
    assert mt.psnr_y(a, a) == float("inf") and abs(mt.ssim(a[..., 0], a[..., 0]) - 1.0) < 1e-12
    # BT.601 studio swing: black -> Y 16, white -> Y 235, grey keeps Cb = Cr = 128
    assert np.allclose(mt.rgb2ycbcr(np.zeros(3)), [16, 128, 128])
    assert np.allclose(mt.rgb2ycbcr(np.full(3, 255.0)), [235, 128, 128], atol=1e-9)
    # constant offset d on all channels -> Y offset d*(sum of the Y row) = d*219/255
    b = a + 3.0
    assert abs(mt.psnr_y(a, b) - 20 * np.log10(255.0 / (3.0 * 219.0 / 255.0))) < 1e-9
    assert abs(mt.psnr_rgb(a / 255, b / 255) - 10 * np.log10(1.0 / (3.0 / 255) ** 2)) < 1e-9
    # AVG_PSNR protocol: borders are excluded, frames are rounded to uint8 first
    vt = rng.random((6, 40, 48, 3))
    vp = vt.copy()
    vp[:, :8] += 0.5
    vp[:2] += 0.5                                   # spatial / temporal borders only
    assert mt.avg_psnr(vt, np.clip(vp, 0, 2)) == float("inf") or mt.avg_psnr(vt, np.clip(vp, 0, 2)) > 100
    assert mt.ssim(a[..., 0], 255 - a[..., 0]) < 0.1
