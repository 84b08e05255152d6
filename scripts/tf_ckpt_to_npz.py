#!/usr/bin/env python
"""Dump every variable of a TF-1.x checkpoint written by a reference script (e.g. algorithm/DeepFM/deepfm.py
--model_dir=...) to an .npz keyed by the TF variable names, for `Estimator.load_variables`:

    python scripts/tf_ckpt_to_npz.py ./model_dir weights.npz        # needs a TensorFlow installation

(TensorFlow is not available in the build container: this script is the reference-side half of the
hand-over and is not exercised by the tests; `Estimator.load_variables` is, with reference-named arrays.)"""
import sys

import numpy as np


def main(model_dir, out):
    import tensorflow as tf
    reader = tf.train.load_checkpoint(model_dir)
    arrays = {name: reader.get_tensor(name) for name in reader.get_variable_to_shape_map()}
    np.savez_compressed(out, **arrays)
    print(f"{len(arrays)} variables -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
