"""Drop-in ``PFNL`` class: same constructor attributes, method names, argument meaning, directory
conventions and prints as the reference's `model/pfnl.py` (class PFNL, :20-332) and
`model/base_model.py` (class VSR, :17-34, save/load :223-243) — with the TF1 graph + session
replaced by one ``PFNLEngine`` (libpfnl_hip on an MI355X).

What is kept:  ``PFNL()`` no-arg ctor and its attributes (:21-37); ``forward(x)`` (:39-80);
``test_video_truth`` (:203-262), ``test_video_lr`` (:264-320), ``testvideos`` (:322-332) incl. the
sliding 7-frame window with clamped indices, ``part``/``num_once`` batching, uint8 quantisation and
the timing print that excludes the first call; ``load`` returning a bool instead of raising
(base_model.py:231-243); ``save``.  ``train``/``eval``/``build`` exist (so `main.py` imports) but
raise NotImplementedError — training is out of scope.
"""
from __future__ import annotations

import glob
import os
import time
from os.path import join
from typing import Dict, Optional

import numpy as np

from . import checkpoint as ckpt
from .spec import PFNLGeometry
from .synth import synthetic_weights


def automkdir(path):                       # reference utils.py:84-86
    if not os.path.exists(path):
        os.makedirs(path)


def imread_rgb(path) -> np.ndarray:
    """RGB uint8 [H,W,3] — what the reference's cv2_imread returns after its BGR->RGB swap
    (utils.py:368-372); cv2 is not available here, PIL reads the same PNG bytes."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def imsave_rgb(path, img) -> None:         # reference utils.py:362-366
    from PIL import Image
    Image.fromarray(np.squeeze(img)).save(path)


def sliding_windows(frames: np.ndarray, num_frames: int) -> np.ndarray:
    """[F,H,W,3] -> [F,T,H,W,3], indices clamped at both ends (model/pfnl.py:238-242)."""
    F = frames.shape[0]
    half = num_frames // 2
    idx = np.clip(np.arange(F)[:, None] + np.arange(-half, half + 1)[None, :], 0, max(F - 1, 0))
    return frames[idx]


def quantise(sr: np.ndarray) -> np.ndarray:
    """model/pfnl.py:254-257: *255, clip to [0,255], np.round (half to even), uint8."""
    return np.round(np.clip(sr * 255.0, 0, 255), 0).astype(np.uint8)


class VSR(object):
    """Attribute/method surface of the reference's base class (model/base_model.py:17-34,223-243)."""

    def __init__(self):
        self.num_frames = 7
        self.scale = 4
        self.in_size = 32
        self.gt_size = self.in_size * self.scale
        self.eval_in_size = [128, 240]
        self.batch_size = 16
        self.eval_basz = 4
        self.learning_rate = 1e-3
        self.end_lr = 1e-4
        self.reload = True
        self.max_step = int(1.5e5 + 1)
        self.decay_step = 1.2e5
        self.train_dir = './data/filelist_train.txt'
        self.eval_dir = './data/filelist_val.txt'
        self.save_dir = './checkpoint'
        self.log_dir = './eval_log.txt'

    # -- engine plumbing (not in the reference; replaces tf.Session / tf.train.Saver) ------------
    _engine = None
    _weights: Optional[Dict[str, np.ndarray]] = None
    device = 0
    num_block = 20

    def geometry(self) -> PFNLGeometry:
        return PFNLGeometry(num_frames=self.num_frames, scale=self.scale, num_block=self.num_block)

    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        """Install variables (TF names -> arrays); the engine is (re)built lazily."""
        self._weights = {k: np.asarray(v, np.float32) for k, v in weights.items()}
        if self._engine is not None:
            self._engine.load_weights(self._weights)

    def _get_engine(self):
        if self._engine is None:
            from .engine import PFNLEngine      # raises if libpfnl_hip.so is missing: no CPU path
            if self._weights is None:
                # the reference runs tf.global_variables_initializer() before (and regardless of)
                # load(): model/pfnl.py:229-232.  Same here: Xavier-uniform random init.
                self._weights = synthetic_weights(self.geometry(), seed=0)
            eng = PFNLEngine(self.geometry(), device=self.device)
            eng.load_weights(self._weights)
            # build-side knob (the reference has none): PFNL_PRECISION=bf16 runs main.py unchanged on the bf16 path of
            # BASELINE.json configs[3]; default = the reference's fp32 arithmetic
            prec = getattr(self, "precision", None) or os.environ.get("PFNL_PRECISION", "fp32")
            if prec != "fp32":
                eng.set_option("precision", prec)
            self._engine = eng
        return self._engine

    def save(self, sess, checkpoint_dir, step):
        """base_model.py:223-229: writes <dir>/VSR-<step>.* and the `checkpoint` state file."""
        if self._weights is None:
            self._weights = synthetic_weights(self.geometry(), seed=0)
        ckpt.save_checkpoint(checkpoint_dir, self._weights, step, model_name="VSR")

    def load(self, sess, checkpoint_dir, step=None):
        """base_model.py:231-243: prints and returns True/False, never raises on a missing checkpoint."""
        print(" [*] Reading SR checkpoints...")
        try:
            found = ckpt.load_checkpoint(checkpoint_dir, self.geometry(), step=step)
        except Exception as e:  # corrupt / unreadable files: the reference would also just fail to restore
            print(" [*] Reading checkpoints... ERROR ({})".format(e))
            return False
        if found is None:
            print(" [*] Reading checkpoints... ERROR")
            return False
        name, weights = found
        self.set_weights(weights)
        print(" [*] Reading checkpoints...{} Success".format(name))
        return True


class PFNL(VSR):
    def __init__(self):
        # reference model/pfnl.py:21-37
        self.num_frames = 7
        self.scale = 4
        self.in_size = 32
        self.gt_size = self.in_size * self.scale
        self.eval_in_size = [128, 240]
        self.batch_size = 16
        self.eval_basz = 4
        self.learning_rate = 1e-3
        self.end_lr = 1e-4
        self.reload = True
        self.max_step = int(1.5e5 + 1)
        self.decay_step = 1.2e5
        self.train_dir = './data/filelist_train.txt'
        self.eval_dir = './data/filelist_val.txt'
        self.save_dir = './checkpoint/pfnl'
        self.log_dir = './pfnl.txt'

    # ---- the hot path -------------------------------------------------------------------------
    def forward(self, x):
        """[B,T,H,W,3] float32 in [0,1] -> [B,1,4H,4W,3] float32, same container type
        (numpy / torch CPU / torch cuda).  Reference: model/pfnl.py:39-80."""
        return self._get_engine().forward(x)

    # ---- out of scope (training) ----------------------------------------------------------------
    def build(self):
        raise NotImplementedError("training graph (model/pfnl.py:82-92) is out of scope of the MI355X inference build")

    def eval(self):
        raise NotImplementedError("validation loop (model/pfnl.py:94-149) is out of scope of the MI355X inference build")

    def train(self):
        raise NotImplementedError("training loop (model/pfnl.py:151-199) is out of scope of the MI355X inference build")

    # ---- inference harness ------------------------------------------------------------------------
    def _run_sequence(self, lrs: np.ndarray, save_path: str, part: int):
        """Shared tail of test_video_truth / test_video_lr (model/pfnl.py:236-262, 293-320)."""
        max_frame = lrs.shape[0]
        if max_frame == 0:
            print('Save at {}'.format(save_path))
            return
        if part > max_frame:
            part = max_frame
        num_once = max_frame // part if max_frame % part == 0 else max_frame // part + 1
        lr_list = sliding_windows(lrs, self.num_frames)
        print('Save at {}'.format(save_path))
        print('{} Inputs With Shape {}'.format(lrs.shape[0], lrs.shape[1:]))
        all_time = []
        for i in range(part):
            batch = lr_list[i * num_once:(i + 1) * num_once]
            if batch.shape[0] == 0:
                break
            st_time = time.time()
            sr = self.forward(np.ascontiguousarray(batch, dtype=np.float32))
            all_time.append(time.time() - st_time)
            for j in range(sr.shape[0]):
                imsave_rgb(join(save_path, '{:0>4}.png'.format(i * num_once + j)), quantise(sr[j][0]))
        all_time = np.array(all_time)
        if max_frame > 0:
            avg = np.mean(all_time[1:]) if len(all_time) > 1 else float('nan')
            print('spent {} s in total and {} s in average'.format(np.sum(all_time), avg))

    def _ensure_loaded(self, reuse):
        if not reuse:
            self.load(None, self.save_dir)

    def test_video_truth(self, path, name='result', reuse=False, part=50):
        """HR pngs in <path>/truth -> blur + decimate (utils.py:169-192) -> SR pngs in <path>/<name>."""
        save_path = join(path, name)
        automkdir(save_path)
        imgs = sorted(glob.glob(join(path, 'truth', '*.png')))
        hr = np.array([imread_rgb(i) for i in imgs]) / 255.
        self._ensure_loaded(reuse)
        if hr.shape[0] == 0:
            return self._run_sequence(np.zeros((0, 0, 0, 3), np.float32), save_path, part)
        # unquantised float LR, as model/pfnl.py:224,234 - on the GPU (pfnl_op_blur_decimate)
        import torch
        from . import ops
        self._get_engine()                       # fails loudly without a device
        dev = "cuda:%d" % self.device
        lrs = ops.blur_decimate(torch.from_numpy(np.ascontiguousarray(hr, np.float32)).to(dev), self.scale).cpu().numpy()
        self._run_sequence(lrs, save_path, part)

    def test_video_lr(self, path, name='result', reuse=False, part=50):
        """LR pngs in <path>/blur{scale} -> SR pngs in <path>/<name>."""
        save_path = join(path, name)
        automkdir(save_path)
        imgs = sorted(glob.glob(join(path, 'blur{}'.format(self.scale), '*.png')))
        lrs = (np.array([imread_rgb(i) for i in imgs]) / 255.).astype(np.float32)
        self._ensure_loaded(reuse)
        if lrs.shape[0] == 0:
            lrs = np.zeros((0, 0, 0, 3), np.float32)
        self._run_sequence(lrs, save_path, part)

    testvideo = test_video_lr      # the README's wording (reference README.md:31)

    def testvideos(self, path='/dev/f/data/video/test2/udm10', start=0, name='pfnl'):
        kind = sorted(glob.glob(join(path, '*')))
        kind = [k for k in kind if os.path.isdir(k)]
        reuse = False
        for idx, k in enumerate(kind):
            if idx >= start:
                if idx > start:
                    reuse = True
                self.test_video_truth(k, name=name, reuse=reuse, part=1000)
