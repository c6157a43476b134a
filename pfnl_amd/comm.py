"""pfnl_comm_* (include/pfnl_hip.h) from Python: RCCL communicators owned by libpfnl_hip, no torch tensors involved.

``torch.distributed`` (or anything else that can move 128 bytes between processes) is only the LAUNCHER: it carries the
RCCL unique id from rank 0 to the other ranks; the weight replica, the scalar reductions and the optional gather then run
on the library's own communicator (SURVEY.md section 8(b),(e)).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import _capi


class Comm:
    """One RCCL communicator (one rank = one process = one GPU)."""

    def __init__(self, nranks: int, rank: int, device: int, unique_id: bytes):
        if len(unique_id) != _capi.COMM_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _capi.COMM_ID_BYTES)
        self._lib = _capi.load_library()
        h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, _capi.COMM_ID_BYTES)
        _capi.check(self._lib.pfnl_comm_init_rank(nranks, rank, buf, device, C.byref(h)))
        self._h = h
        self.rank, self.nranks, self.device = rank, nranks, device

    @classmethod
    def init_all(cls, devices) -> "list[Comm]":
        """One process driving several devices (pfnl_comm_init_all = ncclCommInitAll): communicator i is rank i on devices[i]."""
        lib = _capi.load_library()
        devs = [int(d) for d in devices]
        n = len(devs)
        arr = (C.c_int * n)(*devs)
        hs = (C.c_void_p * n)()
        _capi.check(lib.pfnl_comm_init_all(n, arr, hs))
        out = []
        for i in range(n):
            c = cls.__new__(cls)
            c._lib, c._h = lib, C.c_void_p(hs[i])
            c.rank, c.nranks, c.device = i, n, devs[i]
            out.append(c)
        return out

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
        _capi.check(_capi.load_library().pfnl_comm_get_unique_id(buf))
        return buf.raw

    @classmethod
    def from_exchange(cls, nranks: int, rank: int, device: int,
                      exchange: Callable[[Optional[bytes]], bytes]) -> "Comm":
        """``exchange(id_or_None)`` returns rank 0's id on every rank (e.g. a broadcast_object_list wrapper)."""
        uid = exchange(cls.unique_id() if rank == 0 else None)
        return cls(nranks, rank, device, uid)

    @classmethod
    def from_torch_distributed(cls, device: int) -> "Comm":
        """The id travels through the already initialised torch.distributed group (any backend)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(world, rank, device, box[0])

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pfnl_comm_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def bcast_weights(self, engine, root: int = 0) -> None:
        """Packed device weight blobs of ``engine`` root -> all (non-root engines need no load_weights)."""
        _capi.check(self._lib.pfnl_comm_bcast_weights(self._h, engine._h, root))
        engine._ready = True

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        _capi.check(self._lib.pfnl_comm_allreduce_f64(
            self._h, v.ctypes.data_as(C.POINTER(C.c_double)), v.size, _capi.COMM_SUM if op == "sum" else _capi.COMM_MAX))
        return v

    def rank_and_size(self):
        """(rank, nranks) as the RCCL communicator itself reports them (pfnl_comm_rank -> ncclCommUserRank / ncclCommCount): what a bench
        line quotes to show that N ranks really joined, not what the launcher's environment claimed."""
        r, n = C.c_int(-1), C.c_int(-1)
        _capi.check(self._lib.pfnl_comm_rank(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def barrier(self) -> None:
        _capi.check(self._lib.pfnl_comm_barrier(self._h))

    def allgather(self, local):
        """local: contiguous float32 cuda tensor, same shape on every rank -> [nranks, *shape] cuda tensor."""
        import torch
        out = torch.empty((self.nranks,) + tuple(local.shape), dtype=torch.float32, device=local.device)
        torch.cuda.current_stream(local.device).synchronize()
        _capi.check(self._lib.pfnl_comm_allgather(self._h, C.c_void_p(local.data_ptr()), C.c_void_p(out.data_ptr()),
                                                  local.numel(), None))
        return out
