"""Data-parallel driver pieces: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

PFNL's forward has no cross-batch op (reference model/pfnl.py:44,55: the batch is only the leading
dimension), so clips shard across ranks with NO collective on the data path.  The only exchanges are
(1) one broadcast of the packed weights from rank 0 (12 MB), (2) an optional gather of the SR frames
to rank 0, (3) an all-reduce of a few scalars (squared error, frame count, max time) for PSNR /
throughput.  SURVEY.md §8(e).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from .spec import PFNLGeometry


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n items: the first n % world ranks get one extra."""
    if world < 1 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _dist():
    import torch.distributed as dist
    return dist


def flatten_weights(geom: PFNLGeometry, weights: Dict[str, np.ndarray]) -> np.ndarray:
    return np.concatenate([np.asarray(weights[n], np.float32).ravel() for n, _ in geom.weight_shapes()])


def unflatten_weights(geom: PFNLGeometry, flat: np.ndarray) -> Dict[str, np.ndarray]:
    out, pos = {}, 0
    for name, shape in geom.weight_shapes():
        n = int(np.prod(shape))
        out[name] = np.asarray(flat[pos:pos + n], np.float32).reshape(shape).copy()
        pos += n
    if pos != flat.size:
        raise ValueError("weight blob has {} floats, geometry needs {}".format(flat.size, pos))
    return out


def broadcast_weights(geom: PFNLGeometry, weights: Optional[Dict[str, np.ndarray]], src: int = 0,
                      device: Optional[str] = None) -> Dict[str, np.ndarray]:
    """Rank ``src`` passes its weights, the others pass None; everyone returns the same dict."""
    import torch
    dist = _dist()
    n = geom.num_params()
    if dist.get_rank() == src:
        t = torch.from_numpy(flatten_weights(geom, weights))
    else:
        t = torch.empty(n, dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return unflatten_weights(geom, t.cpu().numpy())


def allreduce_stats(sq_err: float, count: float, seconds: float, device: Optional[str] = None):
    """(sum of squared error, sum of counts, max of seconds) over all ranks."""
    import torch
    dist = _dist()
    s = torch.tensor([sq_err, count], dtype=torch.float64)
    m = torch.tensor([seconds], dtype=torch.float64)
    if device is not None:
        s, m = s.to(device), m.to(device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(s[0]), float(s[1]), float(m[0])


def sharded_forward(forward_fn: Callable[[np.ndarray], np.ndarray], clips: np.ndarray,
                    gather_to: Optional[int] = 0, device: Optional[str] = None) -> Optional[np.ndarray]:
    """Run ``forward_fn`` on this rank's contiguous share of ``clips`` [B,T,H,W,3] (every rank holds the
    same array or at least its own slice) and gather the [B,1,sH,sW,3] result on ``gather_to``
    (None: leave outputs sharded, return the local part).  The gather is a tensor all_gather of equal-size
    (zero-padded) shards - RCCL with backend nccl (``device`` = the rank's GPU), gloo on the host."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    B = clips.shape[0]
    lo, hi = shard_range(B, rank, world)
    local = forward_fn(np.ascontiguousarray(clips[lo:hi])) if hi > lo else None
    if gather_to is None:
        return local
    # every rank needs the frame shape to size its (possibly empty) shard: it is a function of the input alone
    per = max(shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world))
    shape = torch.zeros(4, dtype=torch.int64)
    if local is not None:
        shape = torch.tensor(local.shape[1:], dtype=torch.int64)
    if device is not None:
        shape = shape.to(device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)
    frame = tuple(int(v) for v in shape.cpu())
    pad = torch.zeros((per,) + frame, dtype=torch.float32)
    if local is not None:
        pad[:hi - lo] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
    if device is not None:
        pad = pad.to(device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    if rank != gather_to:
        return None
    outs = []
    for r in range(world):
        a, b = shard_range(B, r, world)
        if b > a:
            outs.append(parts[r][:b - a].cpu().numpy())
    return np.concatenate(outs, axis=0)


def sharded_frame_forward(engine, x, gather_to: Optional[int] = 0):
    """Single-clip multi-GPU sharding: every rank holds the whole clip ``x`` [B,T,H,W,3] (a cuda
    tensor on its GPU; 7 LR frames are small) and produces one horizontal strip of the SR frame with
    ``PFNLEngine.forward_strip`` - the non-local block attends over the whole frame, the trunk recomputes a halo instead of
    exchanging it, so there is NO collective on the data path; the strips are then gathered (equal, zero-padded row
    counts) on ``gather_to`` (None: every rank returns only its rows).  Returns [B,1,sH,sW,3] on ``gather_to``."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    B, T, H, W, _ = x.shape
    s = engine.geom.scale
    # strips of an even number of LR rows (space_to_depth pairs), balanced over the ranks
    lo2, hi2 = shard_range(H // 2, rank, world)
    lo, hi = 2 * lo2, 2 * hi2
    out = torch.zeros(engine.out_shape(B, H, W), dtype=torch.float32, device=x.device)
    if hi > lo:
        engine.forward_strip(x, out, lo, hi - lo)
    torch.cuda.current_stream(x.device).synchronize()
    if gather_to is None:
        return out[:, :, s * lo:s * hi]
    per = 2 * max(shard_range(H // 2, r, world)[1] - shard_range(H // 2, r, world)[0] for r in range(world))
    pad = torch.zeros((B, 1, s * per, s * W, 3), dtype=torch.float32, device=x.device)
    pad[:, :, :s * (hi - lo)] = out[:, :, s * lo:s * hi]
    backend_is_nccl = dist.get_backend() == "nccl"
    send = pad if backend_is_nccl else pad.cpu()
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send)
    if rank != gather_to:
        return None
    rows = []
    for r in range(world):
        a, b = shard_range(H // 2, r, world)
        if b > a:
            rows.append(parts[r][:, :, :s * 2 * (b - a)].to(x.device))
    return torch.cat(rows, dim=2)
