"""PIN KIT - turns "parity unpinned" into a pinned oracle the day somebody has TensorFlow 1.12.

Runs the REFERENCE's own graph (`/root/reference/model/pfnl.py:39-80`, `utils.py:18-71`) under TensorFlow 1.x on the CPU with
the build's seeded synthetic variables assigned BY NAME, and writes inputs, outputs and the reference's variable list to
``tests/golden/tf1_*.npz``.  ``tests/test_tf1_golden.py`` activates as soon as one of these fixtures exists: it checks the
oracle (CPU) and the HIP path (GPU) against the reference's outputs and the variable names / shapes of `pfnl_amd/spec.py`
against what TensorFlow actually created.

This script cannot run in the build image (no TensorFlow, no cv2/h5py - `utils.py:3-6` imports them); it is committed so
that the first person with the reference's environment (Python 3.6, TF 1.12, `README.md:22-23`) can run it:

    cd <checkout of psychopa4/PFNL>            # the reference: its `model/` and `utils.py` must be importable
    python /path/to/repo/tools/make_tf1_golden.py --repo /path/to/repo

Only numeric arrays and the variable-name list are written - nothing of the reference's source.  If `./checkpoint/pfnl`
holds the published checkpoint, `--checkpoint ./checkpoint/pfnl` additionally records the names found in its `.index`.
"""
import argparse
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repo", required=True, help="root of the pfnl_amd repository (fixtures go to <repo>/tests/golden)")
    ap.add_argument("--reference", default=".", help="checkout of psychopa4/PFNL (default: current directory)")
    ap.add_argument("--checkpoint", default=None, help="optional: directory of the published checkpoint (names are recorded)")
    ap.add_argument("--cases", default="small,cfg0", help="comma list of: small (1x7x16x24), cfg0 (1x7x32x32), odd (2x7x20x36)")
    args = ap.parse_args()
    sys.path.insert(0, os.path.abspath(args.repo))
    sys.path.insert(0, os.path.abspath(args.reference))
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")           # the reference's CPU path

    import tensorflow as tf                                      # 1.x
    from model.pfnl import PFNL                                  # the reference's class (imports utils.py: cv2, h5py)
    from pfnl_amd import synth
    from pfnl_amd.spec import PFNLGeometry

    geom = PFNLGeometry()                                        # 7 frames, 4x, 20 blocks: what forward() hard-wires
    weights = synth.synthetic_weights(geom, seed=0)              # name -> HWIO array, names as pfnl_amd/spec.py expects them
    shapes = {"small": (1, 7, 16, 24), "cfg0": (1, 7, 32, 32), "odd": (2, 7, 20, 36)}
    out_dir = os.path.join(os.path.abspath(args.repo), "tests", "golden")
    model = PFNL()
    for case in args.cases.split(","):
        B, T, H, W = shapes[case]
        tf.reset_default_graph()
        x = synth.uniform_clips(B, T, H, W, seed=1234)
        L = tf.placeholder(tf.float32, shape=[B, T, H, W, 3], name="L_test")
        SR = model.forward(L)                                    # model/pfnl.py:39-80
        tf_vars = tf.global_variables()
        names = [v.name for v in tf_vars]                        # e.g. "nlvsr/conv0/kernel:0"
        vshapes = [tuple(int(d) for d in v.shape) for v in tf_vars]
        missing = [n for n in names if n[:-2] not in weights]
        unused = [k for k in weights if k + ":0" not in names]
        cfg = tf.ConfigProto(device_count={"GPU": 0})
        with tf.Session(config=cfg) as sess:
            sess.run(tf.global_variables_initializer())
            for v in tf_vars:
                if v.name[:-2] in weights:
                    v.load(weights[v.name[:-2]].astype(np.float32), sess)
            y = sess.run(SR, feed_dict={L: x})
        dest = os.path.join(out_dir, "tf1_%s.npz" % case)
        np.savez_compressed(dest, x=x.astype(np.float32), y=np.asarray(y, np.float32),
                            var_names=np.array(names), var_shapes=np.array([repr(s) for s in vshapes]),
                            not_in_spec=np.array(missing), not_in_tf=np.array(unused),
                            tf_version=np.array(tf.__version__), weight_seed=np.array(0), input_seed=np.array(1234))
        print("wrote", dest, "y", y.shape, "| TF variables not in spec:", missing, "| spec names TF did not create:", unused)
    if args.checkpoint:
        reader = tf.train.NewCheckpointReader(tf.train.latest_checkpoint(args.checkpoint))
        m = reader.get_variable_to_shape_map()
        dest = os.path.join(out_dir, "tf1_checkpoint_index.npz")
        np.savez_compressed(dest, names=np.array(sorted(m)), shapes=np.array([repr(tuple(m[k])) for k in sorted(m)]))
        print("wrote", dest, len(m), "tensors")


if __name__ == "__main__":
    main()
