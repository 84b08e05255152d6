#!/usr/bin/env python
"""cProfile of the host side of the TFRecord-fed training loop (scripts/bench_tfrecord.py's timed region): where the
milliseconds of a host-bound step go.   python scripts/prof_tfrecord_loop.py [--steps 300]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--examples", type=int, default=65536)
    a = ap.parse_args()
    from recalgorithm_amd import feature_column as fc
    from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn
    from recalgorithm_amd.algorithm.utils import parse_example, train_input_fn
    from recalgorithm_amd.estimator import Estimator, GraphedTrainStep, RunConfig
    from recalgorithm_amd.io import synth
    d = tempfile.mkdtemp(prefix="recalgo_tfrecord_")
    spec = synth.SynthSpec(n_fields=26, max_vocab=1_000_000, seed=9)
    vd, path = d + "/vocabulary/", d + "/train.tfrecord"
    synth.write_vocabularies(spec, vd)
    synth.write_tfrecord(spec, path, a.examples)
    cats = [fc.categorical_column_with_vocabulary_file(n, vd + n + ".txt") for n in spec.names]
    cols = [fc.embedding_column(c, 16) for c in cats]
    labels = [fc.numeric_column("read_comment", default_value=0.0)]

    def parser(serialized):
        f = parse_example(serialized, fc.make_parse_example_spec(cols + labels))
        y = f.pop("read_comment")
        return f, {"read_comment": y}
    parser.columns_getter = lambda: (cols, labels)
    params = {"category_feature_columns": cols, "dense_feature_columns": [], "hidden_units": ["512", "256", "128"],
              "num_cross_layer": 3, "learning_rate": 0.005}
    est = Estimator(dcn_model_fn, params, RunConfig(device="cuda", seed=3))
    it = iter(train_input_fn(path, parser, 4096, None, 10000))
    f, l = est._to_device(*next(it))
    est.build(f, l)
    graphed = GraphedTrainStep(est.train_step, f, l, warmup=2)
    for _ in range(8):
        est.feed_step(graphed, *next(it))
    torch.cuda.synchronize()
    parts = {"next": 0.0, "feed_step": 0.0}
    t_all = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        t0 = time.perf_counter()
        feats, labs = next(it)
        t1 = time.perf_counter()
        est.feed_step(graphed, feats, labs)           # pinned staging -> the graph's input span -> replay
        t2 = time.perf_counter()
        parts["next"] += t1 - t0; parts["feed_step"] += t2 - t1
    pr.disable()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_all
    print("ms/step %.3f" % (dt / a.steps * 1e3), {k: round(v / a.steps * 1e3, 3) for k, v in parts.items()},
          "torch threads", torch.get_num_threads())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print(s.getvalue()[:3500])
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
