"""Checkpoint directory handling with the reference's conventions (model/base_model.py:223-243):

    <dir>/checkpoint                  text state file:  model_checkpoint_path: "VSR-<step>"
    <dir>/VSR-<step>.index / .data-00000-of-00001      TF tensor-bundle (read + written by tfbundle.py)
    <dir>/VSR-<step>.npz                               this build's own fast format, same tensor names

Tensor names are the TF variable names of SURVEY.md §8(a)-W (``nlvsr/conv0/kernel`` ...).  Training
checkpoints also contain optimizer slots (``.../Adam``, ``beta1_power``, ``Variable``); they are
ignored.  Names are matched exactly first, then by suffix (``conv0/kernel``) so that a bundle saved
under a different outer scope still loads; anything unmatched is reported.
"""
from __future__ import annotations

import os
import re
from typing import Dict, Optional, Tuple

import numpy as np

from .spec import PFNLGeometry


def read_state_file(checkpoint_dir: str) -> Optional[str]:
    """Returns the basename in ``model_checkpoint_path`` (tf.train.get_checkpoint_state) or None."""
    path = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.isfile(path):
        return None
    with open(path, "rt") as f:
        for line in f:
            m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"\s*$', line)
            if m:
                return os.path.basename(m.group(1))
    return None


def write_state_file(checkpoint_dir: str, basename: str) -> None:
    path = os.path.join(checkpoint_dir, "checkpoint")
    prev = []
    if os.path.isfile(path):
        with open(path, "rt") as f:
            for line in f:
                m = re.match(r'\s*all_model_checkpoint_paths:\s*"(.*)"\s*$', line)
                if m and m.group(1) != basename:
                    prev.append(m.group(1))
    with open(path, "wt") as f:
        f.write('model_checkpoint_path: "{}"\n'.format(basename))
        for p in prev + [basename]:
            f.write('all_model_checkpoint_paths: "{}"\n'.format(p))


def match_tensors(geom: PFNLGeometry, tensors: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Pick the model's variables out of an arbitrary name->array dict (exact, then suffix match)."""
    out: Dict[str, np.ndarray] = {}
    missing = []
    for name, shape in geom.weight_shapes():
        cand = None
        if name in tensors:
            cand = tensors[name]
        else:
            suffix = name.split("/", 1)[1]
            hits = [k for k in tensors if (k == suffix or k.endswith("/" + suffix)) and
                    tuple(np.shape(tensors[k])) == tuple(shape)]
            if len(hits) == 1:
                cand = tensors[hits[0]]
        if cand is None or tuple(np.shape(cand)) != tuple(shape):
            missing.append(name)
        else:
            out[name] = np.asarray(cand, dtype=np.float32)
    if missing:
        raise KeyError("checkpoint lacks {} of {} tensors, e.g. {}".format(
            len(missing), len(geom.weight_shapes()), missing[:3]))
    # the optional theta / phi projections of the non-local block (nltype 0 checkpoints, reference utils.py:31-42): all four
    # travel with the model, a partial set is an error - never silently dropped (an nltype-0 model would run as nltype 1)
    opt = geom.optional_weight_shapes()
    got = {}
    for name, shape in opt:
        suffix = name.split("/", 1)[1]
        hits = [k for k in tensors if (k == name or k == suffix or k.endswith("/" + suffix)) and tuple(np.shape(tensors[k])) == tuple(shape)]
        if len(hits) == 1:
            got[name] = np.asarray(tensors[hits[0]], dtype=np.float32)
    if got and len(got) != len(opt):
        raise KeyError("checkpoint holds {} of the {} theta / phi tensors of nlblock_0: all or none".format(len(got), len(opt)))
    out.update(got)
    return out


def load_checkpoint(checkpoint_dir: str, geom: PFNLGeometry, step: Optional[int] = None
                    ) -> Optional[Tuple[str, Dict[str, np.ndarray]]]:
    """None if the directory holds no checkpoint (the reference prints ERROR and returns False)."""
    if not os.path.isdir(checkpoint_dir):
        return None
    base = "VSR-{}".format(step) if step is not None else read_state_file(checkpoint_dir)
    if base is None:
        return None
    prefix = os.path.join(checkpoint_dir, base)
    has_npz, has_tf = os.path.isfile(prefix + ".npz"), os.path.isfile(prefix + ".index")
    # The TF bundle is the reference's format and wins; the .npz sibling is this build's cache of it and is only used when it
    # was derived from THIS bundle: it carries the size + sha1 of the .index it was written next to (mtimes say nothing: tar,
    # unzip and rsync -t preserve old ones, so a re-downloaded checkpoint could be shadowed by a newer stale cache).
    tensors = None
    if has_npz:
        with np.load(prefix + ".npz") as z:
            sig = str(z[_SIG_KEY]) if _SIG_KEY in z.files else ""
            if not has_tf or sig == index_signature(prefix):
                tensors = {k: z[k] for k in z.files if k != _SIG_KEY}
    if tensors is not None:
        pass
    elif has_tf:
        from . import tfbundle
        tensors = tfbundle.read_bundle(prefix)
    else:
        return None
    return base, match_tensors(geom, tensors)


_SIG_KEY = "__tf_index_signature__"


def index_signature(prefix: str) -> str:
    """size:sha1 of <prefix>.index ("" when there is none): what ties an .npz cache to the bundle it mirrors."""
    import hashlib
    path = prefix + ".index"
    if not os.path.isfile(path):
        return ""
    with open(path, "rb") as f:
        data = f.read()
    return "{}:{}".format(len(data), hashlib.sha1(data).hexdigest())


def save_checkpoint(checkpoint_dir: str, weights: Dict[str, np.ndarray], step: int, model_name: str = "VSR",
                    fmt: str = "both") -> str:
    """Writes ``<model_name>-<step>`` and points the ``checkpoint`` state file at it (reference model/base_model.py:223-229).
    fmt: "tf" = the reference's tensor bundle, "npz" = this build's cache format, "both" (default) = the bundle plus a cache that
    carries the bundle's signature.  Contract of fmt="npz" next to an EXISTING bundle of the same step (the bundle wins at load
    time unless the cache carries its signature): when the bundle holds exactly these weights the cache is (re)written as its
    mirror - a loop that saves "both" once and "npz" afterwards keeps working; when it holds other weights the call raises
    FileExistsError BEFORE touching the directory (no file written, the state file not moved) - save with fmt="both" to
    rewrite the bundle, or remove it first.  "Mirror" means the MODEL weights: a training bundle's extra tensors (Adam slots,
    global_step, beta powers) are not copied into the cache, which is all an inference load reads; NaNs compare equal (a diverged
    run's weights are mirrored like any other instead of raising)."""
    if fmt not in ("tf", "npz", "both"):
        raise ValueError("fmt must be 'tf', 'npz' or 'both'")
    base = "{}-{}".format(model_name, int(step))
    prefix = os.path.join(checkpoint_dir, base)
    arrs = {k: np.asarray(v, np.float32) for k, v in weights.items()}
    if fmt == "npz" and os.path.isfile(prefix + ".index"):
        from . import tfbundle
        held = tfbundle.read_bundle(prefix)
        same = set(held) >= set(arrs) and all(
            held[k].shape == arrs[k].shape and np.array_equal(np.asarray(held[k], np.float32), arrs[k], equal_nan=True) for k in arrs)
        if not same:
            raise FileExistsError("{}.index holds other weights: save with fmt='both' (rewrites it) or remove the bundle first"
                                  .format(prefix))
    os.makedirs(checkpoint_dir, exist_ok=True)
    if fmt in ("tf", "both"):
        from . import tfbundle
        tfbundle.write_bundle(prefix, arrs)
    if fmt in ("npz", "both"):                                 # written last, with the signature of the bundle it mirrors
        arrs = dict(arrs)
        arrs[_SIG_KEY] = np.array(index_signature(prefix))
        np.savez(prefix + ".npz", **arrs)
    elif os.path.isfile(prefix + ".npz"):
        os.remove(prefix + ".npz")
    write_state_file(checkpoint_dir, base)
    return prefix
