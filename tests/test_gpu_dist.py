"""The N > 1 path with the REAL per-rank compute (PFNLEngine on the GPU) and the RCCL entry points of the C-ABI.

A 1-GPU box cannot give two RCCL ranks a device each (RCCL rejects duplicate devices), so
  * the two-rank tests stack both ranks on the one GPU and move bytes over gloo (the sharding, gather, weight replica and
    bench.py's N>1 branch are exactly the code the 8-GPU run uses; only the transport differs), and
  * pfnl_comm_* is exercised on RCCL itself with a single-rank communicator (every entry point makes its real RCCL call),
    plus a two-rank RCCL test that runs when two devices are visible.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from pfnl_amd import dist as pd  # noqa: E402
from pfnl_amd import synth  # noqa: E402
from pfnl_amd.spec import PFNLGeometry  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pfnl_amd.engine import PFNLEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PFNLGeometry(num_block=2)
        w = pd.broadcast_weights(g, synth.synthetic_weights(g, seed=3) if rank == 0 else None, src=0)
        eng = PFNLEngine(g, device=0)                                  # both ranks on the one visible GPU
        eng.load_weights(w)
        clips = synth.uniform_clips(5, 7, 16, 24, seed=5)              # 5 clips over 2 ranks: 3 + 2
        out = pd.sharded_forward(eng.forward, clips, gather_to=0)
        one = pd.sharded_forward(eng.forward, clips[:1], gather_to=0)  # 1 clip over 2 ranks: rank 1 has an empty shard
        q.put((rank, out, one))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_engine_sharded_equals_unsharded():
    import torch.multiprocessing as mp
    from pfnl_amd.engine import PFNLEngine
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = PFNLGeometry(num_block=2)
    eng = PFNLEngine(g, device=0)
    eng.load_weights(synth.synthetic_weights(g, seed=3))
    clips = synth.uniform_clips(5, 7, 16, 24, seed=5)
    ref = eng.forward(clips)
    assert res[1][1] is None and res[1][2] is None
    assert np.array_equal(res[0][1], ref)                               # sharded == unsharded, bit for bit
    assert np.array_equal(res[0][2], ref[:1])
    eng.close()


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N>1 branch end to end (torch.distributed.run, two ranks, gloo transport): one JSON line, n_gpus 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--no-secondary"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["steps"] == 2 and "cpu_baseline" not in rec


def _bench_two_gloo_ranks(*extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--no-secondary"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_cfg0_weak():
    """north_star's second size (7x32x32 -> 128x128, BASELINE.json configs[0]) on N ranks: one clip per rank, its own roofline object
    and the road the exchanges took named in the line."""
    rec = _bench_two_gloo_ranks("--workload", "cfg0")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["clips_per_gpu"] == 1 and rec["config"]["global_batch"] == 2
    assert "32x32->128x128" in rec["metric"] and "configs[0]" in rec["config"]["workload"]
    assert rec["roofline"] and rec["roofline"]["frac"] > 0 and rec["config"]["conv3x3"] == "small"
    assert rec["config"]["backend"] == "gloo" and rec["config"]["comm"] == "torch.distributed"
    assert abs(rec["value"] - 2 * rec["steps"] / (rec["ms_per_step"] * 1e-3 * rec["steps"])) < 1e-2 * rec["value"]
    rec8 = _bench_two_gloo_ranks("--workload", "cfg0", "--clips-per-gpu", "8")       # stated when overridden
    assert rec8["config"]["clips_per_gpu"] == 8 and rec8["config"]["global_batch"] == 16 and "batch=8" in rec8["config"]["workload"]


def test_bench_two_ranks_strong_scaling():
    """`--strong`: a FIXED global batch in contiguous shards over the ranks - the line says strong, the global batch
    does not grow with N, and value = global batch x steps / max-over-ranks time.  An uneven split (5 clips over 2 ranks) works."""
    rec = _bench_two_gloo_ranks("--strong", "--global-batch", "6")
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 6 and rec["config"]["clips_per_gpu"] == 3
    assert rec["roofline"]["frac"] > 0 and rec["roofline"]["tiles_per_block"] == "5F+2B"
    assert abs(rec["value"] - 6 / (rec["ms_per_step"] * 1e-3)) < 1e-2 * rec["value"]
    rec5 = _bench_two_gloo_ranks("--strong", "--global-batch", "5", "--workload", "cfg0")
    assert rec5["scaling"] == "strong" and rec5["config"]["global_batch"] == 5 and rec5["config"]["clips_per_gpu"] == 3   # rank 0's shard
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--strong", "--global-batch", "6", "--steps", "3", "--warmup", "1",
                          "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert r1["scaling"] == "strong" and r1["n_gpus"] == 1 and r1["config"]["global_batch"] == 6 and r1["config"]["clips_per_gpu"] == 6


def test_bench_self_spawn():
    """`python bench.py --gpus 2` with no launcher around it spawns its own ranks (torch.distributed.run) and still prints
    exactly one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--no-secondary", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 8 and rec["value"] > 0


def _cfg2_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pfnl_amd.engine import PFNLEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PFNLGeometry()                                             # the full 20-block model
        w = pd.broadcast_weights(g, synth.synthetic_weights(g, seed=0) if rank == 0 else None, src=0)
        eng = PFNLEngine(g, device=0)
        eng.load_weights(w)
        clips = synth.uniform_clips(32, 7, 128, 128, seed=32)          # every rank holds the batch; its shard is 4 clips
        lo, hi = pd.shard_range(32, rank, world)
        out = pd.sharded_forward(eng.forward, clips, gather_to=0)
        q.put((rank, (lo, hi), None if out is None else out))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_configs2_eight_ranks_batch32():
    """BASELINE.json configs[2], functionally: B = 32 clips of 7x128x128 sharded over 8 ranks (4 clips each; all eight on the
    one visible GPU, gloo transport - the sharding, weight replica and gather are the code an 8-GPU node runs).  The gathered
    result is bit-equal to the unsharded forward of the same engine, and a sample of it is checked against the oracle."""
    import torch.multiprocessing as mp
    from oracle import pfnl_fast
    from pfnl_amd.engine import PFNLEngine
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg2_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [(4 * i, 4 * i + 4) for i in range(8)]
    got = res[0][2]
    assert got is not None and got.shape == (32, 1, 512, 512, 3) and all(r[2] is None for r in res[1:])
    g = PFNLGeometry()
    w = synth.synthetic_weights(g, seed=0)
    eng = PFNLEngine(g, device=0)
    eng.load_weights(w)
    clips = synth.uniform_clips(32, 7, 128, 128, seed=32)
    for b0 in range(0, 32, 4):                                         # unsharded, in the batches the bench uses
        assert np.array_equal(eng.forward(clips[b0:b0 + 4]), got[b0:b0 + 4])
    eng.close()
    for b in (0, 13, 31):                                              # checker only: three clips, one per "GPU" sampled
        ref = pfnl_fast.FastOracle(w, num_block=20).forward(clips[b:b + 1])
        err = float(np.abs(got[b:b + 1] - ref).max())
        print(f"configs[2] clip {b}: max|hip - oracle| = {err:.3g}")
        assert err < 5e-5, err


def test_bench_rccl_path_with_one_rank():
    """`bench.py`'s N>1 branch on the `nccl` backend - torch.distributed as launcher, the library's own RCCL communicator for the
    weight broadcast and the max-over-ranks time - run with ONE rank (RCCL refuses two ranks on one device; the 8-GPU run is the
    driver's): the line must say which communicator carried it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PFNL_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl"
    assert d["config"]["comm"] == "pfnl_comm (RCCL)", (d["config"], r.stderr[-800:])
    assert d["value"] > 0


def test_weight_replica_without_host_tensors():
    """The RECEIVE side of pfnl_comm_bcast_weights (a handle that never saw pfnl_set_weight: layout built from zeros, the
    sender's device blobs copied over it) cannot run under RCCL on one GPU; pfnl_copy_weights shares that code path.  The
    replica must reproduce the source bit for bit in every arithmetic mode - i.e. nothing data-dependent lives outside the blobs."""
    from pfnl_amd.engine import PFNLEngine
    g = PFNLGeometry(num_block=2)
    src = PFNLEngine(g, device=0)
    src.load_weights(synth.synthetic_weights(g, seed=4))
    dst = PFNLEngine(g, device=0)
    dst.copy_weights_from(src)
    assert dst.missing_weights() == 0
    x = synth.uniform_clips(2, 7, 16, 32, seed=6)
    for opts in ({}, {"conv3x3": "split16", "conv1x1": "split16", "nonlocal": "split16"}, {"conv3x3": "winograd"},
                 {"conv3x3": "direct", "conv1x1": "tiled", "nonlocal": "f32"}, {"precision": "bf16"}):
        for k, v in opts.items():
            src.set_option(k, v)
            dst.set_option(k, v)
        assert np.array_equal(dst.forward(x), src.forward(x)), opts
    other = PFNLEngine(PFNLGeometry(num_block=3), device=0)
    with pytest.raises(Exception):
        other.copy_weights_from(src)                                    # different geometry: refused
    # a received replica holds placeholders, not host tensors: setting ONE tensor afterwards must not silently rebuild the
    # others from zeros - the handle asks for the full set again
    import ctypes as C
    from pfnl_amd import _capi
    k0 = np.ascontiguousarray(synth.synthetic_weights(g, seed=9)["nlvsr/conv0/kernel"])
    _capi.check(dst._lib.pfnl_set_weight(dst._h, b"nlvsr/conv0/kernel", k0.ctypes.data_as(C.c_void_p), (C.c_int64 * 4)(*k0.shape), 4))
    assert dst.missing_weights() == len(g.weight_shapes()) - 1
    with pytest.raises(Exception):
        _capi.check(dst._lib.pfnl_finalize_weights(dst._h))
    dst.load_weights(synth.synthetic_weights(g, seed=4))                # the full set: fine again
    for k, v in {"precision": "fp32", "conv3x3": "auto", "conv1x1": "split16", "nonlocal": "auto"}.items():
        src.set_option(k, v)
        dst.set_option(k, v)
    assert np.array_equal(dst.forward(x), src.forward(x))
    for e in (src, dst, other):
        e.close()


def test_comm_single_rank_rccl():
    """pfnl_comm_* on RCCL itself (one rank): unique id, init, weight broadcast into a handle that never saw
    pfnl_set_weight, scalar reductions, gather, barrier."""
    from pfnl_amd.comm import Comm
    from pfnl_amd.engine import PFNLEngine
    g = PFNLGeometry(num_block=1)
    uid = Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    c = Comm(1, 0, 0, uid)
    eng = PFNLEngine(g, device=0)
    eng.load_weights(synth.synthetic_weights(g, seed=1))
    x = synth.uniform_clips(1, 7, 8, 12, seed=2)
    y0 = eng.forward(x)
    c.bcast_weights(eng, root=0)                                        # root == self: weights unchanged
    assert np.array_equal(eng.forward(x), y0)
    v = c.allreduce([1.5, -2.0, 7.0], "sum")
    assert np.array_equal(v, [1.5, -2.0, 7.0])
    assert np.array_equal(c.allreduce([3.25], "max"), [3.25])
    c.barrier()
    assert c.rank_and_size() == (0, 1)                                  # pfnl_comm_rank: what ncclCommUserRank / ncclCommCount report
    t = torch.arange(24, dtype=torch.float32, device="cuda").reshape(2, 12)
    got = c.allgather(t)
    assert got.shape == (1, 2, 12) and torch.equal(got[0], t)
    with pytest.raises(Exception):
        c.allreduce(np.zeros(65), "sum")                                # n <= 64
    c.close()
    # pfnl_comm_init_all: one process driving a list of devices (here: the one device there is) - ncclCommInitAll
    cs = Comm.init_all([0])
    assert len(cs) == 1 and (cs[0].rank, cs[0].nranks, cs[0].device) == (0, 1, 0) and cs[0].rank_and_size() == (0, 1)
    cs[0].bcast_weights(eng, root=0)
    assert np.array_equal(eng.forward(x), y0)
    assert np.array_equal(cs[0].allreduce([2.0, 5.0], "max"), [2.0, 5.0])
    cs[0].barrier()
    cs[0].close()
    eng.close()


def _rccl_worker(rank, world, uid, q):
    sys.path.insert(0, ROOT)
    from pfnl_amd.comm import Comm
    from pfnl_amd.engine import PFNLEngine
    g = PFNLGeometry(num_block=1)
    torch.cuda.set_device(rank)
    c = Comm(world, rank, rank, uid)
    eng = PFNLEngine(g, device=rank)
    if rank == 0:
        eng.load_weights(synth.synthetic_weights(g, seed=1))
    c.bcast_weights(eng, root=0)                                        # rank 1 never calls pfnl_set_weight
    x = synth.uniform_clips(2, 7, 8, 12, seed=2)
    y = eng.forward(x[rank:rank + 1])
    s = c.allreduce([float(rank + 1), 1.0], "sum")
    m = c.allreduce([0.1 * (rank + 1)], "max")
    q.put((rank, y, s.tolist(), m.tolist()))
    c.close()
    eng.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two RCCL ranks need two devices")
def test_comm_two_ranks_rccl():
    import torch.multiprocessing as mp
    from pfnl_amd.comm import Comm
    from pfnl_amd.engine import PFNLEngine
    uid = Comm.unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, uid, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = PFNLGeometry(num_block=1)
    eng = PFNLEngine(g, device=0)
    eng.load_weights(synth.synthetic_weights(g, seed=1))
    ref = eng.forward(synth.uniform_clips(2, 7, 8, 12, seed=2))
    for r in res:
        assert np.array_equal(r[1], ref[r[0]:r[0] + 1])
        assert r[2] == [3.0, 2.0] and abs(r[3][0] - 0.2) < 1e-12


# ---- single-clip sharding: horizontal strips of one SR frame -------------------------------

@pytest.mark.parametrize("nb,B,H,W,strips,prec", [(3, 1, 96, 64, 4, "fp32"), (2, 2, 40, 36, 3, "fp32"), (20, 1, 128, 128, 2, "fp32"),
                                                    (3, 1, 96, 64, 4, "bf16"), (1, 1, 12, 20, 6, "fp32")])
def test_forward_strips_tile_the_frame(nb, B, H, W, strips, prec):
    """pfnl_forward_strip: the union of the strips (each computed on its own, as a rank would: non-local queries of the strip
    against all keys, trunk on strip + recomputed halo) equals the ORACLE's forward of the whole frame (the build's own full
    forward is only the secondary check: equal up to summation order); rows outside a strip are left untouched."""
    from oracle import pfnl_fast
    from pfnl_amd.engine import PFNLEngine
    geom = PFNLGeometry(num_block=nb)
    eng = PFNLEngine(geom, device=0)
    w = synth.synthetic_weights(geom, seed=0)
    eng.load_weights(w)
    if prec == "bf16":
        eng.set_option("precision", "bf16")
    xh = synth.uniform_clips(B, 7, H, W, seed=nb + H)
    x = torch.from_numpy(xh).cuda()
    ref = pfnl_fast.FastOracle(w, num_block=nb, trunk_dtype="bf16" if prec == "bf16" else "fp32").forward(xh)   # checker only
    full = eng.forward(x)
    out = torch.full_like(full, -7.0)
    bounds = [2 * pd.shard_range(H // 2, r, strips)[0] for r in range(strips)] + [H]
    for r in range(strips):
        lo, hi = bounds[r], bounds[r + 1]
        if hi == lo:
            continue
        before = out.clone()
        eng.forward_strip(x, out, lo, hi - lo)
        torch.cuda.synchronize()
        untouched = torch.ones(4 * H, dtype=torch.bool)
        untouched[4 * lo:4 * hi] = False
        assert torch.equal(out[:, :, untouched], before[:, :, untouched])          # only its own rows
    got = out.cpu().numpy()
    oerr = float(np.abs(got - ref).max())
    opsnr = synth.psnr(got, ref)
    err = (out - full).abs().max().item()
    tol = 2e-6 if prec == "fp32" else 2e-2                                         # bf16 trunk: tile alignment moves rounding points
    print(f"strips nb{nb} {B}x7x{H}x{W} / {strips} ({prec}): max|strips - oracle| = {oerr:.3g} (PSNR {opsnr:.1f} dB), max|strips - full| = {err:.3g}")
    if prec == "fp32":
        assert oerr < 5e-5, oerr                                                   # the bound of tests/test_gpu_forward.py (ABS_TOL)
    else:
        assert opsnr > 60.0, opsnr                                                 # equally-rounded oracle; bf16 rounding points move with tile alignment
    assert err < tol, err
    with pytest.raises(Exception):
        eng.forward_strip(x, out, H - 2, 4)
    eng.close()


def _frame_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pfnl_amd.engine import PFNLEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = PFNLGeometry(num_block=2)
        eng = PFNLEngine(g, device=0)
        eng.load_weights(synth.synthetic_weights(g, seed=3))
        x = torch.from_numpy(synth.uniform_clips(1, 7, 36, 48, seed=9)).cuda()
        y = pd.sharded_frame_forward(eng, x, gather_to=0)
        q.put((rank, None if y is None else y.cpu().numpy()))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_frame():
    """dist.sharded_frame_forward with two ranks (both on the one GPU, gloo transport): rank 0 ends up with the whole SR
    frame, assembled from two independently computed strips."""
    import torch.multiprocessing as mp
    from pfnl_amd.engine import PFNLEngine
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frame_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import pfnl_fast
    ref = pfnl_fast.FastOracle(synth.synthetic_weights(PFNLGeometry(num_block=2), seed=3), num_block=2).forward(
        synth.uniform_clips(1, 7, 36, 48, seed=9))                                   # checker only
    assert res[1][1] is None and res[0][1].shape == ref.shape
    err = float(np.abs(res[0][1] - ref).max())
    print(f"two ranks, one frame: max|gathered strips - oracle| = {err:.3g}")
    assert err < 5e-5, err
