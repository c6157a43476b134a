"""N>1 path on CPU: two gloo processes exercise the sharding, the weight broadcast and the stats
all-reduce that bench.py / the multi-GPU driver use (the GPU run swaps in backend 'nccl' = RCCL).
The per-rank compute is injected (the fp32 oracle here — on the GPU box it is PFNLEngine.forward)."""
import os
import socket
import sys

import numpy as np
import pytest

from pfnl_amd import dist as pd
from pfnl_amd.spec import PFNLGeometry


def test_shard_range_is_a_balanced_partition():
    for n in (0, 1, 4, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [pd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pd.shard_range(4, 2, 2)


def test_flatten_roundtrip():
    from pfnl_amd import synth
    g = PFNLGeometry(num_block=1)
    w = synth.synthetic_weights(g)
    flat = pd.flatten_weights(g, w)
    assert flat.size == g.num_params()
    back = pd.unflatten_weights(g, flat)
    assert all(np.array_equal(back[k], w[k]) for k in w)
    with pytest.raises(ValueError):
        pd.unflatten_weights(g, flat[:-1])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from oracle import pfnl_fast
    from pfnl_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PFNLGeometry(num_block=1)
        w = pd.broadcast_weights(g, synth.synthetic_weights(g, seed=3) if rank == 0 else None, src=0)
        oracle = pfnl_fast.FastOracle(w, num_block=1)
        clips = synth.uniform_clips(3, 7, 8, 12, seed=5)           # 3 clips over 2 ranks: 2 + 1
        out = pd.sharded_forward(oracle.forward, clips, gather_to=0)
        one = pd.sharded_forward(oracle.forward, clips[:1], gather_to=0)   # 1 clip over 2 ranks: rank 1's shard is empty
        assert (one is None) == (rank != 0) and (rank != 0 or np.abs(one - out[:1]).max() < 1e-6)   # (oneDNN: batch-size dependent round-off)
        lo, hi = pd.shard_range(3, rank, world)
        sq, cnt, tmax = pd.allreduce_stats(float(hi - lo), float(hi - lo), 0.1 * (rank + 1))
        q.put((rank, None if out is None else out, sq, cnt, tmax, float(w["nlvsr/conv0/kernel"].sum())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharded_forward():
    import torch.multiprocessing as mp
    from oracle import pfnl_fast
    from pfnl_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = PFNLGeometry(num_block=1)
    w = synth.synthetic_weights(g, seed=3)
    ref = pfnl_fast.FastOracle(w, num_block=1).forward(synth.uniform_clips(3, 7, 8, 12, seed=5))
    assert res[1][1] is None and res[0][1].shape == ref.shape
    assert np.abs(res[0][1] - ref).max() < 1e-6                     # sharded == unsharded
    for r in res:
        assert r[2] == 3.0 and r[3] == 3.0 and abs(r[4] - 0.2) < 1e-12      # SUM, SUM, MAX
        assert abs(r[5] - float(w["nlvsr/conv0/kernel"].sum())) < 1e-6      # broadcast weights
