"""The C-ABI's host side under AddressSanitizer WITHOUT torch (torch's bundled HIP runtime aborts at initialisation under a
preloaded sanitizer runtime; the library's own dependency, /opt/rocm's libamdhip64, does not): numpy containers only, i.e. the
host-pointer form of pfnl_forward (what a `sess.run` replacement uses) - handle creation, weight hand-over,
folding and packing (pfnl_finalize_weights), option changes, workspace sizing, staging copies, the range-flag rerun, weight
replicas, taps, profiling read-out - over the kernel families of all precisions, checked against the oracle on the way.
Run through tools/run_asan.sh --host (sets LD_PRELOAD / PFNL_HIP_LIB)."""
import os
import sys

import numpy as np

os.environ["PFNL_HOST_OUTPUT"] = "pageable"                     # (no torch pinned allocator under the sanitizer: the library's own strips)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pfnl_spec                                   # noqa: E402  (the checker)
from pfnl_amd import _capi, synth                              # noqa: E402
from pfnl_amd.engine import PFNLEngine                         # noqa: E402
from pfnl_amd.spec import PFNLGeometry                         # noqa: E402


def main():
    assert "asan" in _capi.LIB_PATH, "run through tools/run_asan.sh --host"
    n = 0
    # (the third shape moves 1.4 MB in and 3.4 MB out per call: the pinned strips and the copy thread pool of the host-pointer path)
    # (the last shape: 9 clips x 32 chains = 288 (clip, tile) chains = a whole round + 32 chains cut into 7 parts each: round 6's split chains,
    # c10_finalize_kernel, the fused non-local pack (2 048 keys), pfnl_plan / pfnl_get_option)
    for T, scale, nb, B, H, W in ((7, 4, 2, 1, 16, 24), (5, 2, 1, 2, 10, 38), (7, 4, 1, 3, 64, 88), (3, 4, 1, 1, 34, 18), (7, 4, 1, 1, 2, 2), (7, 4, 1, 9, 128, 64)):
        geom = PFNLGeometry(num_frames=T, scale=scale, num_block=nb)
        w = synth.synthetic_weights(geom, seed=T)
        x = synth.uniform_clips(B, T, H, W, seed=H)
        ref = pfnl_spec.forward(x, w, scale=scale, num_block=nb)
        eng = PFNLEngine(geom, device=0)
        eng.load_weights(w)
        for opts in ({}, {"conv3x3": "split16", "small": "off"}, {"conv3x3": "winograd"}, {"conv3x3": "direct", "conv1x1": "tiled"},
                     {"strict_fp32": "on"}, {"small_c10": "off"}, {"split16_sf": "off"}, {"split16_chain": "off"}, {"split16_c10": "off"}, {"graph": "on"}, {"precision": "bf16"},
                     {"precision": "bf16", "bf16_conv10": "separate"}, {"split16_sf0": "on"}, {"split16_splitchains": "off"}, {"split16_mid": "off"}, {"split16_mfma": "32"}, {"precision": "bf16", "bf16_mfma": "32"}):
            if B == 9 and opts.get("conv3x3") in ("winograd", "direct") or (B == 9 and opts.get("strict_fp32")):
                continue                                         # (the f32-MFMA families at the large shape: covered at the small ones, minutes under ASAN)
            for k, v in opts.items():
                eng.set_option(k, v)
                assert eng.get_option(k) == v
            pl = eng.plan(B, H, W)
            assert pl["structure"] and pl["launches_per_block"] >= 2, pl
            if B == 9 and not opts:
                assert pl["structure"] == "chain2_split" and pl["split_parts"] == 7 and pl["nl_pack_fused"] == 1, pl
            for _ in range(2):                                 # (graph=on captures on the second call)
                y = eng.forward(x)
            tol = 3e-2 if opts.get("precision") == "bf16" else 5e-5
            assert y.shape == ref.shape and np.abs(y - ref).max() < tol, (T, opts, np.abs(y - ref).max())
            eng.workspace_bytes(B, H, W)
            eng.tap("nl_out", B, H, W)
            for k in opts:
                eng.set_option(k, {"conv3x3": "auto", "small": "auto", "small_c10": "on", "conv1x1": "split16", "strict_fp32": "off", "split16_sf": "on",
                                   "split16_chain": "on", "split16_c10": "on", "graph": "off", "precision": "fp32", "bf16_nonlocal": "f16",
                                   "bf16_conv10": "fused", "split16_sf0": "off", "split16_splitchains": "auto", "split16_mid": "auto", "split16_mfma": "16", "bf16_mfma": "16"}[k])
            n += 1
        eng.profile(1)
        eng.forward(x)
        eng.profile_read()
        eng.profile(0)
        twin = PFNLEngine(geom, device=0)
        twin.copy_weights_from(eng)
        yt, ye = twin.forward(x), eng.forward(x)
        assert np.array_equal(yt, ye), ("twin != source", (T, scale, nb, B, H, W), float(np.abs(yt - ye).max()), float(np.abs(yt - ref).max()), float(np.abs(ye - ref).max()),
                                        "reruns", twin.range_reruns(), eng.range_reruns(), "repeat", bool(np.array_equal(twin.forward(x), yt)), bool(np.array_equal(eng.forward(x), ye)),
                                        "where", np.argwhere(yt != ye)[:4].tolist(), int((yt != ye).sum()))
        big = x * 3e5                                          # leaves the f16 pipe's domain: the synchronous call reruns on f32 MFMA
        yb = eng.forward(big)
        assert np.isfinite(yb).all() and eng.range_reruns() >= 1
        twin.close()
        eng.close()
    # theta / phi variables and the other arguments of utils.NonLocalBlock
    geom = PFNLGeometry(num_block=1)
    w = synth.synthetic_weights(geom, seed=0)
    rng = np.random.default_rng(6)
    for nme in ("theta/theta", "phi/phi"):
        w[f"nlvsr/nlblock_0/{nme}/kernel"] = (rng.normal(size=(1, 1, 84, 84)) * 0.08).astype(np.float32)
        w[f"nlvsr/nlblock_0/{nme}/bias"] = (0.3 + rng.normal(size=84) * 0.1).astype(np.float32)
    x = synth.uniform_clips(2, 7, 12, 20, seed=4)
    eng = PFNLEngine(geom, device=0)
    eng.load_weights(w)
    for nlt, sub in ((0, 1), (2, 1), (2, 2), (1, 3)):
        eng.set_option("nl_type", str(nlt))
        eng.set_option("nl_sub_sample", str(sub))
        ref = pfnl_spec.forward(x, w, num_block=1, nltype=nlt, sub_sample=sub)
        assert np.abs(eng.forward(x) - ref).max() < 5e-5, (nlt, sub)
        n += 1
    for bad in (("nl_type", "3"), ("conv3x3", "nope"), ("nl_sub_sample", "0")):
        try:
            eng.set_option(*bad)
            raise AssertionError(bad)
        except _capi.PFNLHipError:
            pass
    eng.close()
    print(f"asan host path: {n} configurations, no report")


if __name__ == "__main__":
    main()
