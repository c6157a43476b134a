"""Drop-in ``PFNL`` class: same constructor attributes, method names, argument meaning, directory
conventions and prints as the reference's `model/pfnl.py` (class PFNL, :20-332) and
`model/base_model.py` (class VSR, :17-34, save/load :223-243) — with the TF1 graph + session
replaced by one ``PFNLEngine`` (libpfnl_hip on an MI355X).

What is kept:  ``PFNL()`` no-arg ctor and its attributes (:21-37); ``forward(x)`` (:39-80);
``test_video_truth`` (:203-262), ``test_video_lr`` (:264-320), ``testvideos`` (:322-332) incl. the
sliding 7-frame window with clamped indices, ``part``/``num_once`` batching, uint8 quantisation and
the timing print that excludes the first call; ``load`` returning a bool instead of raising
(base_model.py:231-243); ``save``.  ``eval`` runs the reference's validation protocol (model/pfnl.py:94-149) on the
GPU path; ``train``/``build`` exist (so `main.py` imports) but raise NotImplementedError — training is out of scope
(SURVEY.md §2).

Weights: the reference initialises every variable randomly and then restores a checkpoint if one exists
(model/pfnl.py:229-232); a missing checkpoint therefore yields obvious garbage.  Here a model without weights REFUSES to
run (RuntimeError) unless ``allow_random_init = True`` (or PFNL_ALLOW_RANDOM_INIT=1) opts into a seeded random
initialisation; a checkpoint that exists but cannot be read or does not match the geometry raises, it is never replaced
by anything else.
"""
from __future__ import annotations

import glob
import json
import os
import re
import time
from os.path import join
from typing import Dict, Optional

import numpy as np

from . import checkpoint as ckpt
from .spec import PFNLGeometry
from .synth import synthetic_weights


_STEP_RE = re.compile(r"-(\d+)$")


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
    """PNG, zlib level 1 (what cv2.imwrite - the reference's writer - uses by default in OpenCV 4; PIL's default 6 is ~4x
    slower on a 1080p frame); the decoded pixels are identical at any level."""
    from PIL import Image
    Image.fromarray(np.squeeze(img)).save(path, compress_level=1)


def imread_many(paths, threads: int = 8) -> np.ndarray:
    """[F,H,W,3] uint8 - decoded on a thread pool (PIL releases the GIL while inflating)."""
    if not paths:
        return np.zeros((0, 0, 0, 3), np.uint8)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(paths)))) as pool:
        return np.array(list(pool.map(imread_rgb, paths)))


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
    _engine2 = None                # second handle of the inference harness (two forwards in flight on small batches); made on first use
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
        if getattr(self, "_engine2", None) is not None:      # the harness's second handle (two forwards in flight) follows the first
            self._engine2.copy_weights_from(self._engine)

    allow_random_init = False      # opt-in: run without a checkpoint on a seeded random initialisation
    loaded_step: Optional[int] = None

    def _get_engine(self):
        if self._engine is None:
            from .engine import PFNLEngine      # raises if libpfnl_hip.so is missing: no CPU path
            eng = PFNLEngine(self.geometry(), device=self.device)     # raises without a HIP device
            if self._weights is None:
                # the reference runs tf.global_variables_initializer() before (and regardless of) load():
                # model/pfnl.py:229-232 - its output without a checkpoint is visibly random.  The seeded synthetic set
                # here is scaled to look like a trained model, so it is only used on explicit request.
                if not (self.allow_random_init or os.environ.get("PFNL_ALLOW_RANDOM_INIT") == "1"):
                    eng.close()
                    raise RuntimeError(
                        "PFNL has no weights: no checkpoint was loaded from {!r}; call load()/set_weights(), or set "
                        "allow_random_init = True (PFNL_ALLOW_RANDOM_INIT=1) to run on a seeded random initialisation"
                        .format(self.save_dir))
                self._weights = synthetic_weights(self.geometry(), seed=0)
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
        """base_model.py:231-243: prints and returns True/False; False only when there is NO checkpoint to restore.
        A checkpoint that is present but unreadable / of another geometry raises (as tf.train.Saver.restore does)."""
        print(" [*] Reading SR checkpoints...")
        try:
            found = ckpt.load_checkpoint(checkpoint_dir, self.geometry(), step=step)
        except Exception as e:
            print(" [*] Reading checkpoints... ERROR ({})".format(e))
            raise
        if found is None:
            print(" [*] Reading checkpoints... ERROR")
            return False
        name, weights = found
        self.set_weights(weights)
        m = _STEP_RE.search(name)
        self.loaded_step = int(m.group(1)) if m else None
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
        """The reference's validation protocol (model/pfnl.py:94-149) on the HIP path: every sequence listed in
        ``eval_dir`` (one directory per line, HR frames in ``<dir>/truth/*.png``), centre frames 15, 47, 79, ...,
        clamped 7-frame windows, crop ``[8:8+4*128, 8:8+4*240]``, blur + decimate on the GPU (utils.py:169-192),
        batches of ``eval_basz`` (a trailing partial batch is dropped, as in the reference), RGB MSE of the SR centre
        frame against the HR centre frame, ``PSNR = 10 log10(1/mse)`` per clip; prints the averages and appends one JSON
        line to ``log_dir``.  Returns {"psnr": [...], "mse": [...], "clips": n} (the reference returns None)."""
        print('Evaluating ...')
        if self._weights is None:
            self.load(None, self.save_dir)
        import torch
        from . import ops
        eng = self._get_engine()
        dev = "cuda:%d" % self.device
        border = 8
        in_h, in_w = self.eval_in_size
        out_h, out_w = in_h * self.scale, in_w * self.scale
        T = self.num_frames
        with open(self.eval_dir, 'rt') as f:
            filenames = f.read().splitlines()
        gt_list = [sorted(glob.glob(join(f, 'truth', '*.png'))) for f in filenames if f.strip()]
        center = 15
        batch_gt, batch_cnt, mse_acc = [], 0, []
        for gtlist in gt_list:
            max_frame = len(gtlist)
            for idx0 in range(center, max_frame, 32):
                index = np.clip(np.arange(idx0 - T // 2, idx0 + T // 2 + 1), 0, max_frame - 1).tolist()
                gt = [imread_rgb(gtlist[i]) for i in index]
                gt = [i[border:out_h + border, border:out_w + border, :].astype(np.float32) / 255.0 for i in gt]
                if any(g.shape != (out_h, out_w, 3) for g in gt):
                    raise ValueError("eval frames must be at least {}x{} (crop [{b}:{}+{b}, {b}:{}+{b}]): {}".format(
                        out_h + border, out_w + border, out_h, out_w, gtlist[index[0]], b=border))
                batch_gt.append(np.stack(gt, axis=0))
                if len(batch_gt) == self.eval_basz:
                    bg = np.stack(batch_gt, 0)                                        # [basz,T,out_h,out_w,3]
                    hr = torch.from_numpy(bg.reshape((-1,) + bg.shape[2:])).to(dev)
                    lr = ops.blur_decimate(hr, self.scale).reshape(bg.shape[0], T, in_h, in_w, 3)
                    sr = eng.forward(lr.contiguous()).cpu().numpy()                   # [basz,1,out_h,out_w,3]
                    d = sr.astype(np.float64) - bg[:, T // 2:T // 2 + 1].astype(np.float64)
                    mse_acc.append(np.mean(d * d, axis=(2, 3, 4)))                    # [basz,1]  (model/pfnl.py:90)
                    batch_gt = []
                    print('\tEval batch {} - {} ...'.format(batch_cnt, batch_cnt + self.eval_basz))
                    batch_cnt += self.eval_basz
        if not mse_acc:
            raise ValueError("eval: no full batch of {} clips (sequences need > {} frames)".format(self.eval_basz, center))
        mse_acc = np.concatenate(mse_acc, axis=0)
        psnr_acc = 10 * np.log10(1.0 / mse_acc)
        mse_avg = np.mean(mse_acc, axis=0)
        psnr_avg = np.mean(psnr_acc, axis=0)
        print('Eval PSNR: {}, MSE: {}'.format(psnr_avg, mse_avg))
        with open(self.log_dir, 'a+') as f:
            mse_w = (mse_avg * 1e6).astype(np.int64) / (1e6)
            psnr_w = (psnr_avg * 1e6).astype(np.int64) / (1e6)
            f.write('{' + '"Iter": {} , "PSNR": {}, "MSE": {}'.format(self.loaded_step or 0, psnr_w.tolist(), mse_w.tolist()) + '}\n')
        return {"psnr": psnr_avg.tolist(), "mse": mse_avg.tolist(), "clips": int(mse_acc.shape[0])}

    def train(self):
        raise NotImplementedError("training loop (model/pfnl.py:151-199) is out of scope of the MI355X inference build")

    # ---- inference harness ------------------------------------------------------------------------
    encode_threads = min(16, os.cpu_count() or 1)   # PNG encoders / decoders running while the GPU works on the next batch

    def _run_sequence(self, lrs, save_path: str, part: int):
        """Device-scoped wrapper: the harness helpers (pfnl_op_gather_windows / quantise_u8) launch on the tensor's stream without
        a hipSetDevice of their own, so the whole body runs with this model's device current (self.device != 0)."""
        import torch
        with torch.cuda.device(self.device):
            return self._run_sequence_on_device(lrs, save_path, part)

    def _run_sequence_on_device(self, lrs, save_path: str, part: int):
        """Shared tail of test_video_truth / test_video_lr (model/pfnl.py:236-262, 293-320).  ``lrs`` [F,H,W,3] float32:
        numpy (uploaded ONCE) or already a cuda tensor.  Per batch, all on the device: gather of the clamped T-frame
        windows (pfnl_op_gather_windows), pfnl_forward, uint8 quantisation (pfnl_op_quantise_u8); the uint8 frames
        come back over PCIe (a quarter of the float bytes, pinned double-buffered, asynchronous) and are PNG-encoded on
        worker threads while the NEXT batch runs on the GPU.  The time reported per batch is the device time from the
        window gather to the frames' arrival on the host (HIP events) - what the reference times around sess.run
        (:249-253) -, first batch excluded from the average."""
        max_frame = int(lrs.shape[0])
        if max_frame == 0:
            print('Save at {}'.format(save_path))
            return
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from . import ops
        if part > max_frame:
            part = max_frame
        num_once = max_frame // part if max_frame % part == 0 else max_frame // part + 1
        eng = self._get_engine()
        if torch.is_tensor(lrs):
            frames = lrs.contiguous()
        else:
            frames = torch.from_numpy(np.ascontiguousarray(lrs, dtype=np.float32)).to("cuda:%d" % self.device)
        print('Save at {}'.format(save_path))
        print('{} Inputs With Shape {}'.format(max_frame, tuple(frames.shape[1:])))
        all_time = []
        jobs = []
        stream = torch.cuda.current_stream(frames.device)
        host = []                                                    # pinned uint8 landing buffers: a ring of lanes + 1 (one being drained, `lanes` in flight)
        inflight = []                                                # FIFO of (done event, start event, host buffer, first, count, ran strict, lane)

        # TWO FORWARDS IN FLIGHT (round 6, opt-in: PFNL_HARNESS_INFLIGHT=2).  The reference calls this with part = 50 / 1000 (model/pfnl.py:264,
        # 332), i.e. ONE window per sess.run for every Vid4 / UDM10 sequence (:211-216) - and one clip of that size is the launch structure
        # with the most idle time per pixel (pfnl_plan "mid4" / "small2": ~80 launches of 10 - 30 us).  The batches are independent, so they
        # can alternate between TWO handles on two streams; every forward keeps its batch size and launch structure, so the frames - and the
        # PNG bytes - are those of the serial loop (tests/test_gpu_forward.py::test_harness_two_in_flight_is_byte_identical).  MEASURED
        # (tools/harness_inflight_timing.py, 41 frames of 144x180, profiles/r06_harness_inflight.txt): device time per frame 2.61 -> 2.34 - 2.67 ms
        # (fp32), 1.38 -> 1.12 - 1.38 (bf16) - the gain comes and goes with how the two streams' launches interleave - and the WALL time of
        # the sequence is 5 - 10 % WORSE (a second stream's worth of host work in a loop that is host-bound at this size).  Why it cannot do
        # more: every 3x3 launch fills a CU's LDS with one workgroup, so a second forward's workgroups only start where the first's have
        # exited - what overlaps is launch latency (2 us x 80), not prologues.  Hence not the default; `part` is the lever that works
        # (part = 8: six windows per forward on the "chain2" structure, 0.074 instead of 0.099 us per LR pixel).
        want = os.environ.get("PFNL_HARNESS_INFLIGHT", "auto")
        lanes = 2 if want == "2" and part > 1 else 1                 # ("auto" = 1: see above)
        engines, streams = [eng], [stream]
        if lanes == 2:
            if getattr(self, "_engine2", None) is None:
                self._engine2 = eng.clone()
            else:                                                    # (options may have changed since the clone was made)
                for key in eng.OPTION_KEYS:
                    if self._engine2.get_option(key) != eng.get_option(key):
                        self._engine2.set_option(key, eng.get_option(key))
            s2 = torch.cuda.Stream(device=frames.device)
            s2.wait_stream(stream)                                   # the uploaded frames
            engines.append(self._engine2)
            streams.append(s2)

        # The device-pointer forwards below are asynchronous: the f16-pipe kernels' range fence (include/pfnl_hip.h, "strict_fp32")
        # cannot re-run them by itself.  Once per batch, when its frames have arrived, the flag is read (pfnl_range_flag: no
        # synchronisation of its own); a flagged batch - and the one enqueued behind it before the flag was seen - is computed
        # again on the f32-MFMA kernels, and the rest of the sequence starts there (values beyond binary16's range rarely go away
        # from one batch of a video to the next).  The PNGs never hold quantised non-finite values silently.
        # precision=bf16: `strict_fp32` changes nothing there (its non-local block and conv0 keep their binary16 operands, capi.hip
        # nl_strict), so the recomputation - and the rest of the sequence - runs at precision=fp32 on the strict kernels: the only
        # path of the library that covers the whole fp32 range.  Both options are restored when the sequence ends, however it ends.
        # What is restored is what the LIBRARY held before (pfnl_get_option: a value set by PFNL_STRICT_FP32 / a precision chosen outside this
        # engine's own set_option calls are seen too), and only the keys go_strict changed.
        state = {"strict": False}
        prior = {}                                                   # option -> the library's value before go_strict changed it
        was_bf16 = eng.get_option("precision") == "bf16"

        def go_strict():
            for st in streams:
                st.synchronize()                                     # no forward of the old configuration is in flight when it changes
            if was_bf16:
                prior["precision"] = "bf16"
                print('precision=bf16: a batch left the binary16 range of its non-local block / conv0; recomputed (and the rest of the '
                      'sequence computed) at precision=fp32, strict_fp32=on')
            if eng.get_option("strict_fp32") != "on":
                prior["strict_fp32"] = eng.get_option("strict_fp32")
            for e in engines:
                if was_bf16:
                    e.set_option("precision", "fp32")
                e.set_option("strict_fp32", "on")
            state["strict"] = True

        def recompute_strict(buf, first_, count_, lane_):
            with torch.cuda.stream(streams[lane_]):
                win_ = ops.gather_windows(frames, first_, count_, self.num_frames)
                u8_ = ops.quantise_u8(engines[lane_].forward(win_))
                buf[:count_].copy_(u8_, non_blocking=True)
            streams[lane_].synchronize()
            engines[lane_].range_flagged()                           # (strict path: the fence is not armed; clears a stale flag)

        last_done = [None]

        def drain(item, pool):
            done, started, buf, first_, count_, was_strict, lane_ = item
            done.synchronize()                                       # this batch's frames are on the host
            # device time of the batch: gather + forward + quantise + D2H.  Two in flight: the time from the previous batch's arrival to
            # this one's (the batches overlap: their own spans would count the shared time twice)
            # (= from the later of "its stream reached it" and "the previous batch arrived" to its own arrival: with two forwards in flight
            # the batches overlap and their own spans would count the shared time twice; serial batches: started -> done as before)
            t_own = started.elapsed_time(done)
            t_gap = last_done[0].elapsed_time(done) if last_done[0] is not None else t_own
            all_time.append(max(min(t_own, t_gap), 0.0) * 1e-3)
            last_done[0] = done
            # a flagged batch - and every batch that was enqueued on the f16 pipe before the flag was seen (up to `lanes` of them) - is
            # computed again on the strict kernels
            if not was_strict and (engines[lane_].range_flagged() or state["strict"]):
                if not state["strict"]:
                    go_strict()
                recompute_strict(buf, first_, count_, lane_)
            frames_u8 = buf[:count_].numpy().copy()                  # (the pinned buffer is reused two batches later)
            for j in range(count_):
                jobs.append(pool.submit(imsave_rgb, join(save_path, '{:0>4}.png'.format(first_ + j)), frames_u8[j][0]))

        try:
            with ThreadPoolExecutor(max_workers=max(1, int(self.encode_threads))) as pool:
                for i in range(part):
                    first = i * num_once
                    count = min(num_once, max_frame - first)
                    if count <= 0:
                        break
                    lane = i % lanes
                    with torch.cuda.stream(streams[lane]):
                        started, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        started.record(streams[lane])
                        was_strict = state["strict"]
                        win = ops.gather_windows(frames, first, count, self.num_frames)
                        sr = engines[lane].forward(win)
                        u8 = ops.quantise_u8(sr)
                        k = i % (lanes + 1)
                        if len(host) <= k:
                            host.append(torch.empty((num_once,) + tuple(u8.shape[1:]), dtype=torch.uint8).pin_memory())
                        host[k][:count].copy_(u8, non_blocking=True)
                        done.record(streams[lane])
                    inflight.append((done, started, host[k], first, count, was_strict, lane))
                    if len(inflight) > lanes:                        # while the GPU runs the `lanes` newest batches: the oldest goes to the PNG encoders
                        drain(inflight.pop(0), pool)
                while inflight:
                    drain(inflight.pop(0), pool)
                for j in jobs:
                    j.result()
        finally:
            for st in streams:
                st.synchronize()
            if lanes == 2:
                stream.wait_stream(streams[1])
            if prior:                                                # back to the configuration the library held (also when an encoder raised)
                for e in engines:
                    for key, value in prior.items():
                        e.set_option(key, value)
        all_time = np.array(all_time)
        avg = np.mean(all_time[1:]) if len(all_time) > 1 else float('nan')
        print('spent {} s in total and {} s in average'.format(np.sum(all_time), avg))

    def _ensure_loaded(self, reuse):
        """model/pfnl.py:229-232: restore the checkpoint unless the session is being reused.  load() returning False
        (no checkpoint in save_dir) is tolerated here only if weights were installed some other way - otherwise the
        first forward raises (see _get_engine)."""
        if not reuse:
            if not self.load(None, self.save_dir) and self._weights is None and not (
                    self.allow_random_init or os.environ.get("PFNL_ALLOW_RANDOM_INIT") == "1"):
                raise RuntimeError("no checkpoint in {!r} and no weights installed (set_weights / allow_random_init)"
                                   .format(self.save_dir))

    def test_video_truth(self, path, name='result', reuse=False, part=50):
        """HR pngs in <path>/truth -> blur + decimate (utils.py:169-192) -> SR pngs in <path>/<name>."""
        save_path = join(path, name)
        automkdir(save_path)
        imgs = sorted(glob.glob(join(path, 'truth', '*.png')))
        hr = imread_many(imgs, self.encode_threads) / 255.
        self._ensure_loaded(reuse)
        if hr.shape[0] == 0:
            return self._run_sequence(np.zeros((0, 0, 0, 3), np.float32), save_path, part)
        # unquantised float LR, as model/pfnl.py:224,234 - on the GPU (pfnl_op_blur_decimate)
        import torch
        from . import ops
        self._get_engine()                       # fails loudly without a device
        dev = "cuda:%d" % self.device
        lrs = ops.blur_decimate(torch.from_numpy(np.ascontiguousarray(hr, np.float32)).to(dev), self.scale)   # stays on the device
        self._run_sequence(lrs, save_path, part)

    def test_video_lr(self, path, name='result', reuse=False, part=50):
        """LR pngs in <path>/blur{scale} -> SR pngs in <path>/<name>."""
        save_path = join(path, name)
        automkdir(save_path)
        imgs = sorted(glob.glob(join(path, 'blur{}'.format(self.scale), '*.png')))
        lrs = (imread_many(imgs, self.encode_threads) / 255.).astype(np.float32)
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
