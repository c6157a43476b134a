"""PFNLEngine — the host-side owner of one libpfnl_hip handle (one GPU).

This is the thin Python layer above the C-ABI: shape/dtype validation, weight hand-over, and the
numpy / torch container handling that ``PFNL.forward`` (pfnl_amd/model.py) needs.  All arithmetic
happens in the HIP kernels behind ``pfnl_forward``; nothing here computes.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _capi
from .spec import PFNLGeometry, check_weights


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class _PinnedPool:
    """Free page-locked blocks by size.  Thread-safe; blocks in use are owned by their numpy arrays (a finalizer on the array's base
    returns them here)."""

    def __init__(self):
        import threading
        self._free = {}                       # nbytes -> [ptr, ...]
        self._free_bytes = 0
        # re-entrant: the finalizer (_give) takes the lock too, and the cyclic GC may run it on the thread that already holds it inside
        # array() or _give() (a pinned result caught in a reference cycle)
        self._lock = threading.RLock()

    def _cap(self) -> int:
        return int(float(os.environ.get("PFNL_PINNED_POOL_MB", "256")) * (1 << 20))

    def array(self, shape, nbytes: int) -> Optional[np.ndarray]:
        import weakref
        lib = _capi.load_library()
        ptr = None
        with self._lock:
            lst = self._free.get(nbytes)
            if lst:
                ptr = lst.pop()
                self._free_bytes -= nbytes
        if ptr is None:
            p = C.c_void_p()
            if lib.pfnl_host_alloc(nbytes, C.byref(p)) != 0 or not p.value:
                return None                    # no pinned memory to be had: the caller falls back to pageable memory
            ptr = p.value
        base = (C.c_float * (nbytes // 4)).from_address(ptr)
        fin = weakref.finalize(base, self._give, ptr, nbytes)
        fin.atexit = False                     # at interpreter exit the HIP runtime may be gone already: the OS reclaims the blocks
        return np.ctypeslib.as_array(base).reshape(shape)

    def _give(self, ptr: int, nbytes: int) -> None:
        keep = False
        with self._lock:
            if self._free_bytes + nbytes <= self._cap():
                self._free.setdefault(nbytes, []).append(ptr)
                self._free_bytes += nbytes
                keep = True
        if not keep:
            try:
                _capi.load_library().pfnl_host_free(C.c_void_p(ptr))
            except Exception:                  # interpreter shutdown: the OS takes the memory back
                pass

    def free_bytes(self) -> int:
        return self._free_bytes


_pinned_pool = _PinnedPool()


class PFNLEngine:
    def __init__(self, geom: PFNLGeometry = PFNLGeometry(), device: int = 0):
        self.geom = geom
        self.device = int(device)
        self._lib = _capi.load_library()
        cfg = _capi.pfnl_config(geom.num_frames, geom.scale, geom.mf, geom.num_block, self.device,
                                (C.c_int32 * 3)(0, 0, 0))
        h = C.c_void_p()
        _capi.check(self._lib.pfnl_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._ready = False
        self._options: Dict[str, str] = {}       # what set_option was called with (the library's own view: get_option)

    # ---- lifetime --------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pfnl_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- weights (TF names, HWIO float32) --------------------------------
    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        check_weights(self.geom, weights)
        opt = [(n, s) for n, s in self.geom.optional_weight_shapes() if n in weights]    # theta / phi (nltype 0)
        for name, shape in list(self.geom.weight_shapes()) + opt:
            arr = np.ascontiguousarray(weights[name], dtype=np.float32)
            shp = (C.c_int64 * len(shape))(*shape)
            _capi.check(self._lib.pfnl_set_weight(self._h, name.encode(), arr.ctypes.data_as(C.c_void_p),
                                                  shp, len(shape)))
        _capi.check(self._lib.pfnl_finalize_weights(self._h))
        self._ready = True

    def copy_weights_from(self, other: "PFNLEngine") -> None:
        """Device-to-device replica of ``other``'s packed weights (pfnl_copy_weights): no host tensors, no re-packing."""
        _capi.check(self._lib.pfnl_copy_weights(self._h, other._h))
        self._ready = True

    OPTION_KEYS = ("precision", "strict_fp32", "conv3x3", "conv1x1", "conv2", "merge1", "nonlocal", "nl_type", "nl_sub_sample", "small", "small_c10",
                   "split16_sf", "split16_chain", "split16_c10", "split16_mid", "split16_sf0", "split16_splitchains", "bf16_conv10", "bf16_mfma", "split16_mfma", "graph")

    def clone(self) -> "PFNLEngine":
        """A second handle on the same device with the same weights (device-to-device copy of the packed blobs: pfnl_copy_weights) and the
        same option values as the library holds them now (pfnl_get_option).  Handles are independent: two of them on two streams keep two
        forwards in flight (the harness's small batches, model.py)."""
        other = PFNLEngine(self.geom, device=self.device)
        for k in self.OPTION_KEYS:
            other.set_option(k, self.get_option(k))
        other.copy_weights_from(self)
        return other

    def set_option(self, key: str, value: str) -> None:
        """e.g. ("conv3x3", "auto" | "split16" | "winograd" | "winograd_tile" | "direct"); see include/pfnl_hip.h."""
        _capi.check(self._lib.pfnl_set_option(self._h, key.encode(), value.encode()))
        self._options[key] = value

    def option(self, key: str, default: Optional[str] = None) -> Optional[str]:
        """The value this engine's ``set_option(key, ...)`` last took (None / default: never set, the library's default applies)."""
        return self._options.get(key, default)

    def missing_weights(self) -> int:
        n = C.c_int(0)
        _capi.check(self._lib.pfnl_missing_weights(self._h, C.byref(n)))
        return n.value

    # ---- forward ------------------------------------------------------------------------------
    def _check_input(self, shape, dtype_ok: bool):
        if len(shape) != 5:
            raise ValueError(f"expected [B,T,H,W,3], got shape {tuple(shape)}")
        B, T, H, W, c = (int(s) for s in shape)
        if c != 3 or T != self.geom.num_frames:
            raise ValueError(f"expected [B,{self.geom.num_frames},H,W,3], got {tuple(shape)}")
        if B < 1 or H < 2 or W < 2:
            raise ValueError(f"empty or degenerate input {tuple(shape)}")
        if H % 2 or W % 2:
            raise ValueError("H and W must be even (tf.space_to_depth(2), reference model/pfnl.py:57)")
        if not dtype_ok:
            raise TypeError("input must be float32")
        if not self._ready:
            raise RuntimeError("weights have not been loaded")
        return B, T, H, W

    def out_shape(self, B: int, H: int, W: int):
        s = self.geom.scale
        return (B, 1, s * H, s * W, 3)

    def forward(self, x):
        """x: [B,T,H,W,3] float32 (numpy array, torch CPU tensor or torch tensor on this engine's
        GPU) -> [B,1,sH,sW,3] float32 in the same kind of container (reference model/pfnl.py:39-80).
        Host containers: synchronous (H2D + kernels + D2H inside pfnl_forward).  Device tensors:
        enqueued on torch's current stream."""
        if _is_torch(x):
            import torch
            if x.is_cuda:
                B, T, H, W = self._check_input(x.shape, x.dtype == torch.float32)
                if x.device.index != self.device:
                    raise ValueError(f"tensor on {x.device}, engine on cuda:{self.device}")
                x = x.contiguous()
                out = torch.empty(self.out_shape(B, H, W), dtype=torch.float32, device=x.device)
                stream = torch.cuda.current_stream(x.device).cuda_stream
                _capi.check(self._lib.pfnl_forward(self._h, C.c_void_p(x.data_ptr()), 1,
                                                   C.c_void_p(out.data_ptr()), 1, B, H, W,
                                                   C.c_void_p(stream)))
                return out
            return torch.from_numpy(self.forward(x.detach().numpy()))
        x = np.asarray(x)
        B, T, H, W = self._check_input(x.shape, x.dtype == np.float32)
        x = np.ascontiguousarray(x)
        out = self._host_output(self.out_shape(B, H, W))
        _capi.check(self._lib.pfnl_forward(self._h, x.ctypes.data_as(C.c_void_p), 0,
                                           out.ctypes.data_as(C.c_void_p), 0, B, H, W, None))
        return out

    @staticmethod
    def _host_output(shape) -> np.ndarray:
        """The numpy array a host-pointer forward fills.  Large results are page-locked (pfnl_host_alloc: pfnl_forward then lets the
        copy engine write the result straight into the array instead of staging it through the handle's pinned strip), from a small pool
        owned by this module: a block goes back to the pool when the last view of its array dies, and the pool keeps at most
        PFNL_PINNED_POOL_MB (default 256) of free blocks - anything beyond that is returned to the OS, so many distinct output shapes do
        not accumulate page-locked memory.  No torch involved.  PFNL_HOST_OUTPUT=pageable: plain numpy memory (staged inside the library)."""
        nbytes = int(np.prod(shape)) * 4
        if nbytes >= (512 << 10) and os.environ.get("PFNL_HOST_OUTPUT", "pinned") != "pageable":
            arr = _pinned_pool.array(shape, nbytes)
            if arr is not None:
                return arr
        return np.empty(shape, np.float32)

    @staticmethod
    def host_output_mode() -> str:
        """"pinned" (pool of pfnl_host_alloc blocks) or "pageable": what bench.py records next to its host-pointer line."""
        return "pageable" if os.environ.get("PFNL_HOST_OUTPUT", "pinned") == "pageable" else "pinned (pfnl_host_alloc pool)"

    def forward_device(self, in_ptr: int, out_ptr: int, B: int, H: int, W: int, stream: int = 0) -> None:
        """Raw device-pointer form (asynchronous on ``stream``; 0 = the legacy null stream, i.e. torch's default stream).
        Asynchronous calls cannot re-run on the f32 kernels when the f16-pipe range flag fires: check ``sync()``."""
        self._check_input((B, self.geom.num_frames, H, W, 3), True)
        _capi.check(self._lib.pfnl_forward(self._h, C.c_void_p(in_ptr), 1, C.c_void_p(out_ptr), 1, B, H, W,
                                           C.c_void_p(stream) if stream else None))

    def forward_strip(self, x, out, row0: int, nrows: int) -> None:
        """Single-clip sharding: x [B,T,H,W,3] float32 cuda tensor (the whole clip), out [B,1,sH,sW,3] float32 cuda tensor
        (full size); fills rows [s*row0, s*(row0+nrows)) of ``out`` (pfnl_forward_strip; asynchronous on torch's stream)."""
        import torch
        B, T, H, W = self._check_input(x.shape, x.dtype == torch.float32)
        if not (x.is_cuda and out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == self.out_shape(B, H, W)
                and x.is_contiguous() and out.is_contiguous()):
            raise ValueError("forward_strip needs contiguous float32 cuda tensors, out of the full output shape")
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _capi.check(self._lib.pfnl_forward_strip(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), B, H, W,
                                                 int(row0), int(nrows), C.c_void_p(stream) if stream else None))

    def sync(self) -> None:
        """Synchronise the engine's streams; raises if a device-pointer forward left the f16-pipe kernels' range
        (PFNL_ERR_RANGE, include/pfnl_hip.h "strict_fp32")."""
        _capi.check(self._lib.pfnl_sync(self._h))

    def range_flagged(self) -> bool:
        """Reads and clears the range flag of the device-pointer forwards WITHOUT synchronising (pfnl_range_flag): the caller has
        waited (an event of its own) for the forwards it asks about.  What a pipelined harness checks once per batch."""
        f = C.c_int(0)
        _capi.check(self._lib.pfnl_range_flag(self._h, C.byref(f)))
        return bool(f.value)

    def range_reruns(self) -> int:
        """Host-pointer forwards that were redone on the f32-MFMA kernels (range flag set)."""
        n = C.c_longlong(0)
        _capi.check(self._lib.pfnl_range_reruns(self._h, C.byref(n)))
        return n.value

    def workspace_bytes(self, B: int, H: int, W: int) -> int:
        n = C.c_size_t(0)
        _capi.check(self._lib.pfnl_workspace_bytes(self._h, B, H, W, C.byref(n)))
        return n.value

    def plan(self, B: int, H: int, W: int) -> Dict[str, object]:
        """The launch plan of the progressive-fusion trunk for this shape under the current options (pfnl_plan: the one statement of the
        dispatch rule).  {"structure": "chain2_sf0", "launches_per_block": 2, "conv3x3": "split16", "sf0": 1, ...}."""
        buf = C.create_string_buffer(512)
        _capi.check(self._lib.pfnl_plan(self._h, B, H, W, buf, 512))
        toks = buf.value.decode().split()
        d: Dict[str, object] = {"structure": toks[0]}
        for t in toks[1:]:
            k, v = t.split("=", 1)
            d[k] = int(v) if v.lstrip("-").isdigit() else v
        return d

    def get_option(self, key: str) -> str:
        """The CURRENT value of an option (pfnl_get_option): whatever set it - set_option, the environment at pfnl_create, the default."""
        buf = C.create_string_buffer(64)
        _capi.check(self._lib.pfnl_get_option(self._h, key.encode(), buf, 64))
        return buf.value.decode()

    # ---- measurement / debugging ---------------------------------------------------------------
    def profile(self, enable) -> None:
        """False/0: off; True/1: HIP events around every launch; 2: sampled (PF blocks 3, 13, ... from ten blocks up, every 4th below);
        3: the launches of one PF block only (the dominant classes, 3 events per forward)."""
        _capi.check(self._lib.pfnl_profile_enable(self._h, int(enable)))

    def profile_reset(self) -> None:
        _capi.check(self._lib.pfnl_profile_reset(self._h))

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        n = len(_capi.K_NAMES)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _capi.check(self._lib.pfnl_profile_read(self._h, ms, cnt))
        return {k: {"ms": ms[i], "launches": int(cnt[i])} for i, k in enumerate(_capi.K_NAMES)}

    def tap(self, name: str, B: int, H: int, W: int) -> np.ndarray:
        T = self.geom.num_frames
        shape = {"nl_out": (B, H, W, 3 * T), "trunk": (B, T, H, W, 64), "merge1": (B, H, W, 48)}[name]
        out = np.empty(shape, np.float32)
        _capi.check(self._lib.pfnl_debug_tap(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.size))
        return out
