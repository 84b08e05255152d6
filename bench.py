#!/usr/bin/env python
"""bench.py — CTR examples/sec of one training step (forward + backward + TF1-Adam) of the hot
path on synthetic WeChat-shaped data: 26 sparse fields x emb 16, batch 4096 per GPU.

    python bench.py --gpus N --steps K --warmup W [--model dcn|deepfm|xdeepfm|din]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU over RCCL; per-GPU batch is fixed (weak scaling), `value` is the whole-job
examples/sec = N * batch * K / max-over-ranks(time).  Rank 0 prints ONE JSON line.

Workload at N=1 = BASELINE.json configs[1]: DCN, 3-layer CrossNet on the [B, 416] gathered
embeddings + MLP 512,256,128, B = 4096, fp32, TF1-Adam on every variable (dense Adam over the
whole embedding arena, the reference's semantics).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector == fp32 MFMA peak


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="dcn", choices=["dcn", "deepfm", "xdeepfm", "din", "fibinet", "pnn", "fwfm", "nfm", "afm", "ffm"])
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--emb", type=int, default=16)
    ap.add_argument("--max-vocab", type=int, default=1_000_000)
    ap.add_argument("--data-batches", type=int, default=32, help="distinct synthetic batches rotated through")
    ap.add_argument("--no-tunable", dest="tunable", action="store_false",
                    help="keep hipBLASLt's default fp32 GEMM selection for the context MLP (default: PyTorch "
                         "TunableOp picks the GEMM kernels during warm-up; selections are frozen before timing)")
    ap.add_argument("--capacity-factor", type=float, default=0.75,
                    help="N > 1: bucket capacity of the static id/row exchange, in units of requests / world.  The buckets "
                         "hold DISTINCT rows: on this bench's Zipf batches 30 %% of the requests, spread evenly over the "
                         "owners (max fill 0.30 at N = 2..8), so 0.75 leaves 2.5 x head room and halves the all_to_all "
                         "bytes of the library default (1.5); an overflow repeats the run with the capacity doubled")
    ap.add_argument("--big-table-rows", type=int, default=0,
                    help="replace the vocabulary of the last field by a table of this many rows "
                         "(BASELINE configs[4]: one 100M x 16 table)")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE.json configs[4]: DeepFM with ONE 100 M-row x emb16 table among its fields, rows sharded r %% N "
                         "over the N GPUs (RCCL all_to_all for the id buckets and rows), 4096 examples per GPU (global batch "
                         "32 768 at N = 8); = --model deepfm --big-table-rows 100000000")
    ap.add_argument("--lazy-adam", action="store_true",
                    help="LazyAdamOptimizer on the embedding tables (rows without a gradient keep weights and moments): a "
                         "labelled DEVIATION from the reference's tf.train.AdamOptimizer (SURVEY.md §8f-1)")
    ap.add_argument("--sweep-batches", type=int, default=2048,
                    help="optimizer-state sweep after the timed run: this many FRESH batches (one per step, never repeated; "
                         "drawn on the device — models with history / dense inputs use the host generator and at most 192) "
                         "and a forced all-rows-live run; 0 = off")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--allow-eager", action="store_true",
                    help="N > 1: if the hipGraph capture of the step (RCCL all_to_alls included) fails, time eager launches "
                         "instead of failing the run (the line then says launch = eager)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true",
                    help="skip the `host_fed` leg (TFRecord bytes -> vocabulary ids -> training step, scripts/bench_tfrecord.py on a "
                         "bounded file: ~25 s)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dropout-rate", type=float, default=0.0,
                    help="training-mode tf.layers.dropout rate of the models that have one (DeepFM / DIN / FiBiNET / PNN / NFM; the "
                         "reference scripts default to 0.1, deepfm.py:39).  Default 0 = BASELINE.json's parity configuration; a run "
                         "with a rate > 0 says so in config.workload")
    ap.add_argument("--no-extra-models", action="store_true",
                    help="default run (--model dcn, N = 1): skip the DeepFM / xDeepFM / DIN lines appended under `models`")
    args = ap.parse_args(argv)
    if args.config5:
        args.model = "deepfm"
        args.big_table_rows = args.big_table_rows or 100_000_000
    return args


def self_launch(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): re-exec under
    torch.distributed.run, one rank per GPU of this node, rendezvous on 127.0.0.1 — so the command is self-contained and
    the line it prints says "n_gpus": N because N ranks really ran.  Fails loudly when the node has fewer than N GPUs
    (RECALGO_DIST_BACKEND=gloo_staged, the bring-up mode in which ranks share devices, is exempt)."""
    import socket
    staged = os.environ.get("RECALGO_DIST_BACKEND", "nccl") == "gloo_staged"
    have = torch.cuda.device_count()
    if have < args.gpus and not staged:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} HIP device(s); refusing to print a line for "
                         f"{args.gpus} GPUs measured on fewer (set RECALGO_DIST_BACKEND=gloo_staged for the functional "
                         "bring-up mode, whose numbers are not benchmark results)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs between processes on this stack
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(sys.executable, cmd, env)


def build_estimator(args, device, rank=0, world=1, before_build=None):
    from recalgorithm_amd import feature_column as fc
    from recalgorithm_amd.estimator import Estimator, RunConfig
    from recalgorithm_amd.io import synth

    spec = synth.SynthSpec(n_fields=args.fields, max_vocab=args.max_vocab,
                           with_history=(args.model == "din"), history_len=50 if args.model == "din" else None,
                           with_dense=args.model in ("nfm", "afm"))
    if args.big_table_rows:
        spec.vocabs[-1] = int(args.big_table_rows)
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    hidden = ["512", "256", "128"]
    if args.model == "dcn":
        from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn as model_fn
        params = {"category_feature_columns": [fc.embedding_column(c, args.emb) for c in cats],
                  "dense_feature_columns": [], "hidden_units": hidden, "num_cross_layer": 3,
                  "learning_rate": 0.005}
        workload = f"DCN 3-layer CrossNet + MLP 512,256,128; {args.fields} fields x emb{args.emb}; batch {args.batch}/GPU"
    elif args.model == "deepfm":
        from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn as model_fn
        # DeepFM consumes its columns in the order given (one input_layer call per column, deepfm.py:187-190);
        # listing them in the order of the resident id matrix (sorted names, as fc.input_layer sorts for
        # the other models) lets the fused kernel read the matrix in place
        cats = sorted(cats, key=lambda c: c.key)
        params = {"first_order_feature_columns": [fc.indicator_column(c) for c in cats],
                  "second_order_feature_columns": [fc.embedding_column(c, args.emb) for c in cats],
                  "hidden_units": hidden, "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005}
        workload = f"DeepFM FM1+FM2+MLP 512,256,128 (BN); {args.fields} fields x emb{args.emb}; batch {args.batch}/GPU"
    elif args.model == "xdeepfm":
        from recalgorithm_amd.algorithm.xDeepFM.xdeepfm import xdeepfm_model_fn as model_fn
        params = {"category_feature_columns": [fc.embedding_column(c, args.emb) for c in cats],
                  "dense_feature_columns": [], "hidden_units": hidden, "learning_rate": 0.005,
                  "embedding_dim": args.emb, "cin_layer_feature_maps": ["128", "128"]}
        workload = f"xDeepFM CIN [128,128] + MLP 512,256,128; {args.fields} fields x emb{args.emb}; batch {args.batch}/GPU"
    elif args.model == "din":
        # DIN: 25 profile fields x emb + the target feedid and its 50-long history sharing one table
        from recalgorithm_amd.algorithm.DIN.din import din_model_fn as model_fn
        cmap = dict(zip(spec.names, cats))
        his = fc.categorical_column_with_identity("his_read_comment_7d_seq", cmap["feedid"].num_buckets)
        his.is_sequence = True
        feed = cmap.pop("feedid")
        feed.is_sequence = True
        shared = fc.shared_embedding_columns([feed, his], args.emb, combiner="mean")
        params = {"dense_feature_columns": [], "category_feature_columns": [fc.embedding_column(c, args.emb) for c in cmap.values()],
                  "target_feedid_feature_columns": [shared[0]], "sequence_feature_columns": [shared[1]],
                  "hidden_units": hidden, "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005,
                  "activation": "dice", "mini_batch_aware_regularization": True, "l2_lambda": 0.2,
                  "use_softmax": False, "sequence_max_length": 50}
        workload = (f"DIN attention over a 50-long history (default non-softmax branch, dice) + MLP 512,256,128; "
                    f"{args.fields - 1} profile fields + target feed x emb{args.emb}; batch {args.batch}/GPU")
    elif args.model == "fibinet":
        from recalgorithm_amd.algorithm.FiBiNET.fibinet import fibinet_model_fn as model_fn
        params = {"category_feature_columns": [fc.embedding_column(c, args.emb) for c in cats],
                  "dense_feature_columns": [], "hidden_units": hidden, "dropout_rate": 0.0, "batch_norm": True,
                  "learning_rate": 0.005, "embedding_dim": args.emb, "reduction_ratio": 2,
                  "bilinear_interaction_type": "all"}
        workload = f"FiBiNET SENET(r=2) + bilinear 'all' + MLP 512,256,128; {args.fields} fields x emb{args.emb}; batch {args.batch}/GPU"
    elif args.model == "pnn":
        from recalgorithm_amd.algorithm.PNN.pnn import pnn_model_fn as model_fn
        params = {"category_feature_columns": [fc.embedding_column(c, args.emb) for c in cats],
                  "hidden_units": hidden, "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005,
                  "output_dimension": 1024, "product_method": "IPNN", "weight_regularizer": 0.0,
                  "embedding_dim": args.emb}
        workload = f"PNN IPNN D=1024 + MLP 512,256,128; {args.fields} fields x emb{args.emb}; batch {args.batch}/GPU"
    elif args.model == "fwfm":              # SURVEY.md §8f-3 sibling: no MLP, the step is the sparse path + Adam
        from recalgorithm_amd.algorithm.FwFM.fwfm import fwfm_model_fn as model_fn
        cats = sorted(cats, key=lambda c: c.key)
        params = {"first_order_feature_columns": [fc.indicator_column(c) for c in cats],
                  "second_order_feature_columns": [fc.embedding_column(c, args.emb) for c in cats],
                  "embedding_dim": args.emb, "learning_rate": 0.005}
        workload = f"FwFM first order + field-pair-weighted second order; {args.fields} fields x emb{args.emb}; batch {args.batch}/GPU"
    elif args.model in ("nfm", "afm"):      # §8f-3 siblings; both add a linear term over the 16 dense features
        from recalgorithm_amd.algorithm._common import DENSE_FEATURES
        params = {"dense_feature_columns": [fc.numeric_column(k, default_value=0.0) for k in DENSE_FEATURES],
                  "category_feature_columns": [fc.embedding_column(c, args.emb) for c in cats], "learning_rate": 0.005}
        if args.model == "nfm":
            from recalgorithm_amd.algorithm.NFM.nfm import nfm_model_fn as model_fn
            params.update({"hidden_units": hidden, "dropout_rate": 0.0, "batch_norm": True})
            workload = (f"NFM bi-interaction pooling + BN + dropout 0.1 + MLP 512,256,128 (BN); {args.fields} fields x "
                        f"emb{args.emb}; batch {args.batch}/GPU")
        else:
            from recalgorithm_amd.algorithm.AFM.afm import afm_model_fn as model_fn
            params.update({"embedding_dim": args.emb, "attention_factor": 128})
            workload = (f"AFM pair Hadamard products ({args.fields * (args.fields - 1) // 2} pairs) + attention net 128; "
                        f"{args.fields} fields x emb{args.emb}; batch {args.batch}/GPU")
    elif args.model == "ffm":               # §8f-3 sibling: F-1 sub-tables per field
        from recalgorithm_amd.algorithm.FFM.ffm import ffm_model_fn as model_fn
        cols = [fc.indicator_column(c) for c in cats]
        params = {"one_hot_category_feature_columns": cols, "embedding_dim": args.emb, "learning_rate": 0.005,
                  "fields_vocabulary_size_tuple": [(c.key, c.num_buckets) for c in cats]}
        workload = (f"FFM first order + field-aware pair dots ({args.fields} fields x {args.fields - 1} sub-tables x "
                    f"emb{args.emb}); batch {args.batch}/GPU")
    else:
        raise SystemExit(f"--model {args.model}: unknown")
    if args.lazy_adam:
        params["lazy_adam"] = True
    if args.big_table_rows:
        workload += (f"; one {int(args.big_table_rows):,}-row table among the fields"
                     + (f", rows sharded r % {world} (BASELINE.json configs[4])" if world > 1 else " (BASELINE.json configs[4] on one GPU)"))
    if getattr(args, "dropout_rate", 0.0) > 0 and "dropout_rate" in params:
        params["dropout_rate"] = float(args.dropout_rate)
        workload += f"; dropout {args.dropout_rate:g} (hash-keyed masks, csrc/dropout.h)"
    est = Estimator(model_fn=model_fn, params=params, config=RunConfig(device=device, seed=42))
    feats, labels, _ = synth.device_features(spec, args.batch, device, batch_index=rank)
    if before_build is not None:
        before_build(est)
    est.build(feats, labels)
    return est, spec, feats, labels, workload


def box_sanity(device):
    """Which box is this?  The pool's boxes differ (one ran every HBM-bound kernel 30-60 % slower): a device
    copy rate measured here, and the clocks / power cap rocm-smi reports, let a reader tell a slow box from a slow tree."""
    out = {}
    try:
        x = torch.empty(1 << 28, dtype=torch.uint8, device=device)          # 256 MiB
        y = torch.empty_like(x)
        y.copy_(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y.copy_(x)
        e1.record()
        e1.synchronize()
        out["hbm_copy_GBs"] = round(10 * 2 * x.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)     # read + write
        del x, y
    except Exception as e:
        out["hbm_copy_error"] = f"{type(e).__name__}: {e}"
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--json"], capture_output=True, text=True,
                           timeout=20)
        js = json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else {}
        card = js.get(f"card{device.index or 0}") or (next(iter(js.values())) if js else {})
        for k, v in card.items():
            lk = k.lower()
            if "sclk" in lk and "level" in lk:
                out["sclk"] = v
            elif "mclk" in lk and "level" in lk:
                out["mclk"] = v
            elif "max graphics package power" in lk:
                out["power_cap_W"] = v
            elif "graphics package power" in lk or "average" in lk and "power" in lk:
                out["power_W"] = v
    except Exception as e:
        out["rocm_smi"] = f"unavailable ({type(e).__name__})"
    return out


def event_time_ms(fn, reps=20, replays=10):
    """Average duration of ONE `fn()` launch, from HIP events recorded on the stream the kernels
    run on.  `reps` launches are captured into a hipGraph and the graph is replayed `replays`
    times between the two events, so that the ~5-10 us of Python/ctypes launch overhead per call
    is not mistaken for kernel time (the kernels of this path run 5-25 us at batch 4096)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    g.replay()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(replays):
        g.replay()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / (reps * replays)


def kernel_rooflines(args, est, feats, device):
    """Per-kernel average launch time (HIP events) and algorithmic-bytes roofline fraction for
    the hand-written kernels of the step (SURVEY.md §8d byte model)."""
    import ctypes
    from recalgorithm_amd import _lib
    lib = _lib.load()
    B, F, K = args.batch, args.fields, args.emb
    d = F * K
    class _Cur:                       # the stream current at call time (the capture stream)
        @property
        def _as_parameter_(self):
            return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    st = _Cur()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    store = est.store
    ar = next((a for a in store.arenas.values() if a.K == K), next(iter(store.arenas.values())))
    names = sorted(feats.keys())
    ids = torch.stack([feats[n] for n in names if isinstance(feats[n], torch.Tensor) and feats[n].dtype == torch.int64], 1).contiguous()
    rb = torch.tensor([ar.tables[t][0] for t in list(ar.tables)[:F]], dtype=torch.int64, device=device)
    sd = getattr(ar, "sharding", None)
    if sd is not None:
        # N > 1: this rank's arena holds the rows r % N == rank at r // N.  The per-kernel table times
        # the local kernels on the shard, so the batch's rows are folded into it (same distribution)
        rows = ids + rb.unsqueeze(0)
        ids = torch.where(ids >= 0, torch.div(rows, sd.sh.world, rounding_mode="floor"), ids).contiguous()
        rb = torch.zeros_like(rb)
    x0 = torch.empty(B, d, device=device)
    out = torch.empty(B, d, device=device)
    g = torch.randn(B, d, device=device)
    res = []

    def add(name, fn, alg_bytes, flops=0.0):
        ms = event_time_ms(fn, reps=20 if flops < 1e9 else 4, replays=10 if flops < 1e9 else 3)
        t_hbm, t_fl = alg_bytes / (HBM_PEAK_GBS * 1e9), flops / (FP32_PEAK_TFLOPS * 1e12)
        r = {"kernel": name, "avg_us": round(ms * 1e3, 3), "alg_bytes": int(alg_bytes),
             "achieved_GBs": round(alg_bytes / (ms * 1e-3) / 1e9, 1),
             "bound": "mfma" if t_fl > t_hbm else "hbm",
             "frac": round(max(t_hbm, t_fl) / (ms * 1e-3), 4)}
        if flops:
            r["alg_flops"] = float(flops)
            r["achieved_TFLOPs"] = round(flops / (ms * 1e-3) / 1e12, 2)
        res.append(r)

    add("gather_fwd", lambda: lib.recalgo_embedding_gather_fwd(p(ids), p(ar.weight), p(rb), B, F, K, p(x0), d, 0, st),
        B * (F * 8 + 2 * d * 4))
    from recalgorithm_amd import sparse as sp
    owner = sp.plan_of(ar) is not None
    if not owner:
        add("gather_bwd", lambda: lib.recalgo_embedding_gather_bwd(p(ids), p(g), p(rb), B, F, K, d, 0, p(ar.grad), None, st),
            B * (F * 8 + 2 * d * 4))
        ar.grad.zero_()
    else:
        # owner-computes scatter fused with the optimizer (csrc/sparse.hip), THREE launches per step for a one-lookup model:
        # `prepare` with the lookup (bucket counts of its requests + catch-up of their lagging rows + this step's share of
        # the deferred-Adam sweep, one grid), then `place` (entries into their buckets, a tile's duplicates summed) and `apply`
        # (per-row sums in request order, TF1 Adam on the owned rows) after the backward pass; the prefix of the bucket
        # counts rides on the optimizer's dense launch.  Timed on a scratch copy of the arena's state in the state the timed
        # steps left it in; lr = 0; the step counter does not advance, so the replay loops of the deferred Adam have nothing
        # to do here (the in-step rocprofv3 averages — `in_step`, below — include them)
        import copy
        sc = copy.copy(ar)
        sc.weight, sc.m, sc.v, sc.grad = ar.weight.clone(), ar.m.clone(), ar.v.clone(), ar.grad
        pl0 = sp.plan_of(ar)
        sc.sparse = sp.ArenaPlan(sc)
        if pl0.last_step is not None:
            sc.sparse.last_step, sc.sparse.lr_ring = pl0.last_step.clone(), pl0.lr_ring.clone()
        sc.sparse.betas = pl0.betas
        lazy = pl0.last_step is None
        step_dev = store.opt_state["step"]
        distinct = int(torch.unique((ids + rb.unsqueeze(0))[ids >= 0]).numel())

        def prep_only():
            with torch.enable_grad():
                sp.begin_lookup(sc, store, ids, None, rb, 0, B, F)
            sc.sparse.sources = []                  # (the counts stay: they are never consumed by this loop)

        def sparse_step():
            with torch.enable_grad():
                src = sp.begin_lookup(sc, store, ids, None, rb, 0, B, F)
            src.set_grad(g)
            sp.apply(sc, lazy, step_dev, 0.0, 0.9, 0.999, 1e-8)
        sparse_step()
        n_req = B * F
        rows_sweep = 0 if lazy else -(-ar.weight.shape[0] // sp.sweep_period())
        alg_prep = n_req * 8 + distinct * 4 + rows_sweep * 4                     # ids, a counter word per distinct row, last_step of the sweep share
        alg_apply = n_req * (8 + 8 + 8) + n_req * K * 4 + distinct * (6 * K * 4 + 8)    # ids + keys out/in, gradient rows, (w, m, v) in/out + last_step
        add("sparse_prepare(counts + catch-up + sweep share: the lookup's one launch; nothing lags in this loop)", prep_only, alg_prep)
        res[-1]["part_of"] = "sparse_step"       # (timed again inside the three-launch sequence below)
        sc.sparse.counted = None
        add("sparse_step(prepare + place + apply: row sums, Adam on owned rows)", sparse_step, alg_prep + alg_apply)
        res[-1]["distinct_rows"] = distinct
        res[-1]["requests"] = n_req
        res[-1]["launches"] = 3                  # a SEQUENCE of launches (per-kernel times: `in_step`): never `roofline`
        del sc
    if args.model == "dcn":
        L = 3
        w = torch.randn(L, d, device=device) * 0.05
        b = torch.randn(L, d, device=device) * 0.05
        dw, db, dx0 = torch.empty_like(w), torch.empty_like(b), torch.empty_like(x0)
        ws = torch.empty(lib.recalgo_cross_bwd_workspace_bytes(B, d, L), dtype=torch.uint8, device=device)
        add("cross_fwd", lambda: lib.recalgo_cross_fwd(p(x0), d, p(w), p(b), B, d, L, p(out), d, st), B * 2 * d * 4)
        if sd is None:
            # the step runs the gather INSIDE the cross stack's forward launch (ops.gather_feeds_cross): the two separate
            # kernels above stay in the table for reference, the composite counts the fused launch
            res[-1]["part_of"] = "gather_cross_fwd"
            next(r for r in res if r["kernel"] == "gather_fwd")["part_of"] = "gather_cross_fwd"
            add("gather_cross_fwd(embedding gather + 3 cross layers, x0 written on the way)",
                lambda: lib.recalgo_gather_cross_fwd(p(ids), p(ar.weight), p(rb), B, F, K, p(w), p(b), L, p(x0), d, p(out), d, st),
                B * (F * 8 + 3 * d * 4))
        add("cross_bwd", lambda: lib.recalgo_cross_bwd(p(x0), d, p(w), p(b), p(g), d, None, B, d, L, p(dx0), p(dw), p(db), p(ws), 0, st),
            B * 3 * d * 4)
        cross_row = res[-1]
        cross_ops = (x0, w, b, g, dx0, ws, L)
    if args.model == "xdeepfm":
        m, D = F, K
        x3 = torch.randn(B, m, D, device=device)
        ws = torch.empty(lib.recalgo_cin_layer_bwd_workspace_bytes(B, m, 128, 128, D), dtype=torch.uint8, device=device)
        for li, Hk in enumerate((m, 128)):
            N = 128
            xk = torch.randn(B, Hk, D, device=device)
            w = torch.randn(Hk * m, N, device=device) * 0.02
            o = torch.empty(B, N, D, device=device)
            pool = torch.empty(B, N, device=device)
            go = torch.randn(B, N, D, device=device)
            dx0, dxk, dw = torch.empty_like(x3), torch.empty_like(xk), torch.empty_like(w)
            fl = 2.0 * B * D * Hk * m * N
            byt = B * (m * D + Hk * D + N * D + N) * 4 + w.numel() * 4
            add(f"cin_fwd(L{li + 1},Hk={Hk})", lambda: lib.recalgo_cin_layer_fwd(p(x3), p(xk), p(w), B, m, Hk, N, D, p(o), p(pool), N, 0, st),
                byt, fl)
            add(f"cin_bwd(L{li + 1},Hk={Hk})", lambda: lib.recalgo_cin_layer_bwd(p(x3), p(xk), p(w), p(go), p(pool), N, 0, B, m, Hk, N, D,
                                                                                 p(dx0), 0, p(dxk), 0, p(dw), p(ws), st),
                2 * byt, 2.0 * fl)      # bwd = 2x fwd (one GEMM G.W^T feeding dX^k and dX^0, one for dW)
    if args.model in ("deepfm", "fwfm"):
        w1 = next(a for n, a in store.arenas.items() if n.endswith("_w1"))
        bias = torch.zeros(1, device=device)
        emb = torch.empty(B, d, device=device)
        fm1, fm2, fs = torch.empty(B, device=device), torch.empty(B, device=device), torch.empty(B, K, device=device)
        g1, g2 = torch.randn(B, device=device), torch.randn(B, device=device)
        add("deepfm_sparse_fwd", lambda: lib.recalgo_deepfm_sparse_fwd(p(ids), p(ar.weight), p(w1.weight), p(bias), p(rb), B, F, K,
                                                                      p(emb), p(fm1), p(fm2), p(fs), st),
            B * (F * 8 + F * K * 4 + F * 4 + F * K * 4 + 8))                       # 3648 B/example
        # (backward: the FM second-order gradient g_emb + g_fm2 * (S - e) is formed where `place` / `apply` load the gradient
        # rows — sparse_step above — and recalgo_deepfm_sparse_bwd is not part of the step)
        ar.grad.zero_(); w1.grad.zero_()
    if args.model == "din":
        T, H = 50, K
        q = torch.randn(B, H, device=device)
        keys = torch.randn(B, T, H, device=device)
        kl = torch.full((B,), T, dtype=torch.int32, device=device)
        f1w, f1b = torch.randn(4 * H, 64, device=device) * 0.1, torch.zeros(64, device=device)
        f2w, f2b = torch.randn(64, 32, device=device) * 0.1, torch.zeros(32, device=device)
        f3w, f3b = torch.randn(32, 1, device=device) * 0.1, torch.zeros(1, device=device)
        o = torch.empty(B, H, device=device)
        go = torch.randn(B, H, device=device)
        dq, dk = torch.empty_like(q), torch.empty_like(keys)
        dws = [torch.empty_like(t) for t in (f1w, f1b, f2w, f2b, f3w, f3b)]
        ws = torch.empty(lib.recalgo_din_attention_bwd_workspace_bytes(B, T, H), dtype=torch.uint8, device=device)
        fl = 2.0 * B * T * (4 * H * 64 + 64 * 32 + 32) + 2.0 * B * T * H            # MLP + weighted sum, as the reference computes it
        byt = B * ((T + 1) * H * 4 + 4 + H * 4)
        add("din_attention_fwd", lambda: lib.recalgo_din_attention_fwd(p(q), p(keys), p(kl), p(f1w), p(f1b), p(f2w), p(f2b), p(f3w),
                                                                      p(f3b), B, T, H, 0, p(o), st), byt, fl)
        res[-1]["prof"] = ["din16::fwd_kernel", 0]
        add("din_attention_bwd", lambda: lib.recalgo_din_attention_bwd(p(q), p(keys), p(kl), p(f1w), p(f1b), p(f2w), p(f2b), p(f3w),
                                                                      p(f3b), p(go), B, T, H, 0, p(dq), p(dk), *[p(t) for t in dws],
                                                                      p(ws), st), 2 * byt + B * T * H * 4, 2.0 * fl)       # bwd = 2x fwd (the kernel's recompute of the forward is not useful work)
        res[-1]["prof"] = ["din16::bwd_kernel", 0]
        vals = torch.randint(0, 1000, (B * T,), device=device)
        offs = torch.arange(0, B * T + 1, T, device=device, dtype=torch.int64)
        so, sl = torch.empty(B, T, H, device=device), torch.empty(B, dtype=torch.int32, device=device)
        add("sequence_gather_fwd", lambda: lib.recalgo_sequence_gather_fwd(p(vals), p(offs), p(ar.weight), B, T, H, p(so), p(sl), st),
            B * T * (8 + 2 * H * 4))
        # (backward: a ragged source of the arena's scatter plan — sparse_step — not recalgo_sequence_gather_bwd)
        ar.grad.zero_()
    if args.model == "fibinet":
        E = torch.randn(B, F, K, device=device)
        V = torch.empty_like(E)
        Rd = K // 2
        w1s, w2s = torch.randn(F, Rd, device=device) * 0.3, torch.randn(Rd, F, device=device) * 0.3
        Wo, Wsn = torch.randn(K, K, device=device) * 0.2, torch.randn(K, K, device=device) * 0.2
        P_ = (F - 1) * (F - 2) // 2
        out2 = torch.empty(B, P_, 2 * K, device=device)
        g2 = torch.randn(B, P_, 2 * K, device=device)
        dE, dV = torch.empty_like(E), torch.empty_like(E)
        dWo, dWs, dw1, dw2 = torch.empty_like(Wo), torch.empty_like(Wsn), torch.empty_like(w1s), torch.empty_like(w2s)
        wsb = torch.empty(lib.recalgo_bilinear_bwd_workspace_bytes(B, F, K, 2, 0), dtype=torch.uint8, device=device)
        wss = torch.empty(lib.recalgo_senet_bwd_workspace_bytes(B, F, K, Rd), dtype=torch.uint8, device=device)
        add("senet_fwd", lambda: lib.recalgo_senet_fwd(p(E), p(w1s), p(w2s), B, F, K, Rd, p(V), None, st), B * 2 * F * K * 4)
        add("senet_bwd", lambda: lib.recalgo_senet_bwd(p(E), p(w1s), p(w2s), p(E), B, F, K, Rd, p(dE), 0, p(dw1), p(dw2), p(wss), st),
            B * 3 * F * K * 4)
        add("bilinear_fwd(all, 2 sets)", lambda: lib.recalgo_bilinear_fwd(p(E), p(Wo), p(V), p(Wsn), B, F, K, 0, p(out2), 2 * K, 0, st),
            B * (2 * F * K * 4 + P_ * 2 * K * 4))
        add("bilinear_bwd(all, 2 sets)", lambda: lib.recalgo_bilinear_bwd(p(E), p(Wo), p(V), p(Wsn), p(g2), 2 * K, 0, B, F, K, 0, p(dE),
                                                                         p(dWo), p(dV), p(dWs), p(wsb), st),
            B * (4 * F * K * 4 + P_ * 2 * K * 4))
    if args.model in ("pnn", "fwfm"):
        E = torch.randn(B, F * K, device=device)
        T_ = F * (F + 1) // 2
        D_ = 1024
        phi, dphi = torch.empty(B, T_, device=device), torch.randn(B, T_, device=device)
        th = torch.randn(D_, F, device=device) * 0.1
        om, dom = torch.empty(T_, D_, device=device), torch.randn(T_, D_, device=device)
        dE, dth = torch.empty_like(E), torch.empty_like(th)
        add("pnn_features_fwd(IPNN)", lambda: lib.recalgo_pnn_features_fwd(p(E), B, F, K, 0, p(phi), T_, st), B * (F * K + T_) * 4)
        add("pnn_features_bwd(IPNN)", lambda: lib.recalgo_pnn_features_bwd(p(E), p(dphi), T_, B, F, K, 0, p(dE), 0, st),
            B * (2 * F * K + T_) * 4)
        if args.model == "pnn":
            add("pnn_weights_fwd(IPNN)", lambda: lib.recalgo_pnn_weights_fwd(p(th), D_, F, K, 0, p(om), st), (D_ * F + T_ * D_) * 4)
            add("pnn_weights_bwd(IPNN)", lambda: lib.recalgo_pnn_weights_bwd(p(th), p(dom), D_, F, K, 0, p(dth), st),
                (2 * D_ * F + T_ * D_) * 4)
    if args.model in ("nfm", "afm", "ffm"):          # §8f-3 sibling kernels (csrc/siblings.hip)
        e = torch.randn(B, F, K, device=device)
        if args.model == "nfm":
            o, go, de = torch.empty(B, K, device=device), torch.randn(B, K, device=device), torch.empty_like(e)
            add("bi_interaction_fwd", lambda: lib.recalgo_bi_interaction_fwd(p(e), B, F, K, p(o), st), B * (F * K + K) * 4)
            add("bi_interaction_bwd", lambda: lib.recalgo_bi_interaction_bwd(p(e), p(go), B, F, K, p(de), st),
                B * (2 * F * K + K) * 4)
        elif args.model == "afm":
            P_ = F * (F - 1) // 2
            pr, att = torch.randn(B, P_, K, device=device), torch.randn(B, P_, device=device)
            o, sc = torch.empty(B, K, device=device), torch.empty(B, P_, device=device)
            go, dpr, datt = torch.randn(B, K, device=device), torch.empty_like(pr), torch.empty_like(att)
            add("attention_pool_fwd", lambda: lib.recalgo_attention_pool_fwd(p(pr), p(att), B, P_, K, p(o), p(sc), st),
                B * (P_ * K + 2 * P_ + K) * 4)
            add("attention_pool_bwd", lambda: lib.recalgo_attention_pool_bwd(p(pr), p(sc), p(go), B, P_, K, p(dpr), p(datt), st),
                B * (2 * P_ * K + 2 * P_ + K) * 4)
        else:
            xf = torch.randn(B, F, F - 1, K, device=device)
            o, go, dxf = torch.empty(B, device=device), torch.randn(B, device=device), torch.empty_like(xf)
            add("ffm_pairs_fwd", lambda: lib.recalgo_ffm_pairs_fwd(p(xf), B, F, K, p(o), st), B * (F * (F - 1) * K + 1) * 4)
            add("ffm_pairs_bwd", lambda: lib.recalgo_ffm_pairs_bwd(p(xf), p(go), B, F, K, p(dxf), st),
                B * (2 * F * (F - 1) * K + 1) * 4)
    # the fused logit head + sigmoid + cross-entropy + head backward launch (csrc/mlp.hip) on this model's head shape
    # (DCN: concat[cross out d, dnn out 128]; the others: the last hidden layer) and the step's one input copy
    import ctypes as _ct
    head_parts = [torch.randn(B, d, device=device), torch.randn(B, 128, device=device)] if args.model == "dcn" else \
        [torch.randn(B, 128, device=device)]
    hw = [torch.randn(t.shape[1], device=device) * 0.05 for t in head_parts]
    hb = torch.zeros(1, device=device)
    lbl = (torch.rand(B, device=device) < 0.04).float()
    Ch = sum(t.shape[1] for t in head_parts)
    rows_p = int(lib.recalgo_logit_loss_partial_rows(B))
    partials = torch.empty(rows_p, Ch + 2, device=device)
    lg, pr, dl = torch.empty(B, 1, device=device), torch.empty(B, 1, device=device), torch.empty(B, 1, device=device)
    hdx = [torch.empty_like(t) for t in head_parts]
    ptrs = lambda ts: (_ct.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    wi = (_ct.c_int * len(head_parts))(*[t.shape[1] for t in head_parts])
    pa, pw, pdx = ptrs(head_parts), ptrs(hw), ptrs(hdx)
    # (DCN's step runs head + loss inside the fused last-layer launch, below: the row stays in the table for reference only)
    fused_tail = args.model == "dcn" and bool(lib.recalgo_tail_dense_head_supported(256, 128, d))
    add("logit_loss(head + sigmoid + CE + head backward)",
        lambda: lib.recalgo_logit_loss_fwd_bwd(pa, pw, wi, len(head_parts), p(hb), None, None, p(lbl), None, B, 1.0, p(lg), p(pr),
                                               p(dl), pdx, None, p(partials), st), B * (2 * Ch * 4 + 16) + rows_p * (Ch + 2) * 4)
    if fused_tail:
        res[-1]["part_of"] = "tail_dense_head"
    src_b, dst_b = torch.empty(B * F * 8 + B * 4, dtype=torch.uint8, device=device), torch.empty(B * F * 8 + B * 4, dtype=torch.uint8, device=device)
    add("input_copy(batch -> the graph's static buffers)",
        lambda: lib.recalgo_copy_bytes(p(dst_b), p(src_b), src_b.numel(), st), 2 * src_b.numel())
    # context MLP on the fp32 matrix cores (csrc/dense.hip): forward (bias + ReLU fused) and the merged backward launch
    # (input + weight gradient tiles, ReLU mask and bias gradient fused) of THIS MODEL's layers (dense_layer_table): the
    # three layers d_in -> 512 -> 256 -> 128, behind PNN's product layer [416 (+) 351] -> 1024 and FiBiNET's 9600 -> 512
    # library GEMM (SURVEY.md 8d rows K6 / K7+K8 / "MLP")
    from recalgorithm_amd import ops
    keep, slab_bytes = [], 0
    d_in, first_kind = dense_layer_table(args, d)
    if first_kind == "pnn":
        # PNN's D-way contraction relu(emb W + phi Omega + b) (pnn.py:133-181): ONE two-operand-pair forward launch, TWO merged
        # backward launches (one per operand pair).  Roofline = SURVEY 8d K6: 1.74 MFLOP per example forward (lz 0.85 M +
        # lp 0.885 M AS THE REFERENCE COMPUTES IT — the Gram form contracts 351 instead of 432 features per unit), bwd = 2x
        D_, Tp = 1024, F * (F + 1) // 2
        T4 = (Tp + 3) // 4 * 4
        xe, ph = torch.randn(B, d, device=device), torch.randn(B, T4, device=device)
        we, om = torch.randn(d, D_, device=device) / d ** 0.5, torch.randn(T4, D_, device=device) / T4 ** 0.5
        bp = torch.zeros(D_, device=device)
        yp = ops.dense_fwd(xe, we, bp, True, x2=ph, w2=om)
        gp = torch.randn(B, D_, device=device) * (torch.rand(B, D_, device=device) > 0.5)
        dwe, dom_, dbp = torch.empty_like(we), torch.empty_like(om), torch.empty_like(bp)
        fl_ref = 1.74e6 * B
        byt = B * (d + D_) * 4 + (d + T4) * D_ * 4
        add("pnn_product_fwd([416 (+) 351 Gram features] -> 1024, bias + ReLU; one launch)",
            lambda: ops.dense_fwd(xe, we, bp, True, x2=ph, w2=om), byt, fl_ref)
        res[-1]["engine_flops"] = 2.0 * B * (d + T4) * D_

        def pnn_bwd():
            ops.dense_bwd(ph, gp, yp, om, dom_, None)
            ops.dense_bwd(xe, gp, yp, we, dwe, dbp, defer=True)
            ops._dense_pending.clear()
        def pnn_bwd_phi():
            ops.dense_bwd(ph, gp, yp, om, dom_, None)

        def pnn_bwd_emb():
            ops.dense_bwd(xe, gp, yp, we, dwe, dbp, defer=True)
            ops._dense_pending.clear()
        res[-1]["prof"] = ["dense_fwd_kernel", 0]
        add("pnn_product_bwd:emb pair (d emb, d linear_w, d bias; ReLU mask staged)", pnn_bwd_emb,
            (B * (2 * d + 2 * D_) + 2 * d * D_) * 4, 2.0 * 0.85e6 * B)
        res[-1].update(part_of="pnn_product_bwd", prof=["dense_bwd_kernel<true, true>", 0], engine_flops=4.0 * B * d * D_)
        add("pnn_product_bwd:phi pair (d phi, d omega; ReLU mask staged)", pnn_bwd_phi,
            (B * (2 * T4 + 2 * D_) + 2 * T4 * D_) * 4, 2.0 * 0.885e6 * B)
        res[-1].update(part_of="pnn_product_bwd", prof=["dense_bwd_kernel<true, true>", 1], engine_flops=4.0 * B * T4 * D_)
        add("pnn_product_bwd(two merged dgrad + wgrad launches, one per operand pair)", pnn_bwd, 2 * byt, 2.0 * fl_ref)
        res[-1]["launches"] = 2
        res[-1]["engine_flops"] = 4.0 * B * (d + T4) * D_
        keep.append((xe, ph, we, om, yp, gp, dwe, dom_))
    elif first_kind == "library":
        # FiBiNET's [B, 300, 32] -> flatten 9600 -> dense 512 (fibinet.py:177-199): hipBLASLt through torch (nn._mfma_dense),
        # forward (bias + ReLU epilogue), weight gradient, input gradient: three library launches + the mask / bias pass
        Kl, Nl = d_in[0], 512
        xl = torch.randn(B, Kl, device=device)
        wl = torch.randn(Kl, Nl, device=device) / Kl ** 0.5
        bl = torch.zeros(Nl, device=device)
        gl = torch.randn(B, Nl, device=device) * (torch.rand(B, Nl, device=device) > 0.5)
        dwl, dxl = torch.empty_like(wl), torch.empty_like(xl)
        fl = 2.0 * B * Kl * Nl
        add(f"library_gemm_fwd({Kl}->{Nl}, hipBLASLt)", lambda: torch._addmm_activation(bl, xl, wl), (B * (Kl + Nl) + Kl * Nl) * 4, fl)
        add(f"library_gemm_wgrad({Kl}->{Nl}, hipBLASLt)", lambda: torch.mm(xl.t(), gl, out=dwl), (B * (Kl + Nl) + Kl * Nl) * 4, fl)
        add(f"library_gemm_dgrad({Kl}->{Nl}, hipBLASLt)", lambda: torch.mm(gl, wl.t(), out=dxl), (B * (Kl + Nl) + Kl * Nl) * 4, fl)
        keep.append((xl, wl, bl, gl, dwl, dxl))
        d_in = d_in[1:]
    widths = d_in
    for li in range(len(widths) - 1):
        Kd, Nd = widths[li], widths[li + 1]
        xd = torch.randn(B, Kd, device=device)
        wd = torch.randn(Kd, Nd, device=device) / Kd ** 0.5
        bd = torch.zeros(Nd, device=device)
        # (the gradient wrt a ReLU layer's output arrives masked by [y > 0] — about half of it zeros, as in the step: the
        # matrix cores' clock follows the data, MI355X_MICROARCH.md "DVFS give-back"; recorded per row as `grad_zero_fraction`)
        gd = torch.randn(B, Nd, device=device) * (torch.rand(B, Nd, device=device) > 0.5)
        yd = ops.dense_fwd(xd, wd, bd, True)
        dwd, dbd = torch.empty_like(wd), torch.empty_like(bd)
        fl = 2.0 * B * Kd * Nd
        last_fused = fused_tail and li == len(widths) - 2
        add(f"dense_fwd({Kd}->{Nd})", lambda: ops.dense_fwd(xd, wd, bd, True), (B * (Kd + Nd) + Kd * Nd) * 4, fl)
        res[-1]["prof"] = ["dense_fwd_kernel", li + (1 if first_kind == "pnn" else 0)]
        if last_fused:
            # DCN's last hidden layer runs inside ONE launch with the head, the loss and the backward of all three down to
            # the layer's input (csrc/tailfuse.hip, dcn.py:166-172); its weight gradient is a launch of its own.  The separate
            # forward / merged-backward rows of this layer stay in the table for reference (`part_of`).
            res[-1].pop("prof")
            res[-1]["part_of"] = "tail_dense_head"
            xd.clamp_(min=0)
            side_t, wh_t = torch.randn(B, d, device=device), torch.randn(d + Nd, device=device) * 0.05
            rows_t = int(lib.recalgo_tail_partial_rows(B))
            part_t = torch.empty(rows_t, d + Nd + 2, device=device)
            dsd_t, dz_t, dh_t = torch.empty_like(side_t), torch.empty(B, Nd, device=device), torch.empty_like(xd)
            add(f"tail_dense_head(dense {Kd}->{Nd} + head over [{d} | {Nd}] + sigmoid-CE + backward to the layer's input)",
                lambda: lib.recalgo_tail_dense_head_fwd_bwd(p(xd), Kd, p(wd), p(bd), Nd, p(side_t), d, 1, p(wh_t), p(wh_t[d:]), p(hb),
                                                            p(lbl), None, B, 1.0, p(lg), p(pr), p(dl), p(dsd_t), p(dz_t), p(dh_t),
                                                            p(part_t), st),
                B * (2 * Kd + 2 * d + Nd + 16) * 4 + Kd * Nd * 4 + rows_t * (d + Nd + 2) * 4, 2.0 * fl)
            res[-1]["prof"] = ["tail_dense_head_kernel", 0]
            dwd, dbd = torch.empty_like(wd), torch.empty_like(bd)

            n_pend = len(ops._dense_pending)

            def wgrad_once():
                ops.dense_bwd_weights(xd, dz_t, None, dwd, dbd, defer=True)
                del ops._dense_pending[n_pend:]
            add(f"dense_wgrad({Kd}->{Nd}: the fused layer's weight gradient, slabs summed by the deferred-sum launch)", wgrad_once,
                (B * (Kd + Nd) + Kd * Nd) * 4, fl)
            res[-1]["part_of"] = "dense_bwd_rider"            # (in the step it rides in the launch of the layer below's backward)
            ops.dense_bwd_weights(xd, dz_t, None, dwd, dbd, defer=True)       # leaves the layer's split slabs + its pending entry
            keep.append((xd, wd, bd, side_t, wh_t, part_t, dsd_t, dz_t, dh_t, dwd, dbd))
            slab_bytes += int(lib.recalgo_dense_bwd_weights_workspace_bytes(B, Kd, Nd)) + (Kd * Nd + Nd) * 4
        # (the ONE merged launch, as in the step: the fixed-order sum of the batch-split slabs is a job of the step's
        #  deferred-sum launch, listed below — not a second launch per layer)
        # As in the step (nn.ReluSource): the gradient a layer receives was masked with its ReLU output by the kernel that
        # produced it (the layer above / the loss tail), so no mask is staged; a layer whose input is itself a ReLU output
        # (all but the first of a plain stack; every layer behind PNN's / FiBiNET's first one) masks the input gradient it writes.
        pm = xd.clamp_(min=0) if (li > 0 or first_kind is not None) else None
        n_pend = len(ops._dense_pending)            # (the layers before this one keep their entries for the deferred-sum row)

        def bwd_once():
            ops.dense_bwd(xd, gd, None, wd, dwd, dbd, defer=True, premask=pm)
            del ops._dense_pending[n_pend:]
        add(f"dense_bwd({Kd}->{Nd})", bwd_once, (B * (2 * Kd + 2 * Nd + (Kd if pm is not None else 0)) + 2 * Kd * Nd) * 4, 2.0 * fl)
        res[-1]["prof"] = ["dense_bwd_kernel<true, false>", li]
        if last_fused:
            res[-1].pop("prof")
            res[-1]["part_of"] = "tail_dense_head"
            continue
        if fused_tail and li == len(widths) - 3 and li > 0 and lib.recalgo_dense_bwd_cross_rider_supported(d, cross_ops[6]):
            # In DCN's step this layer's merged backward launch also carries, as RIDERS dispatched in front of its own tiles, the
            # weight gradient of the fused last layer and the CrossNet backward (recalgo_dense_bwd_rider): timed here as the ONE
            # launch the step runs; the three separate rows stay in the table for reference (`part_of`)
            res[-1].pop("prof")
            res[-1]["part_of"] = "dense_bwd_rider"
            cross_row["part_of"] = "dense_bwd_rider"
            Kn, Nn = widths[li + 1], widths[li + 2]
            rx, rg = torch.randn(B, Kn, device=device).clamp_(min=0), torch.randn(B, Nn, device=device) * (torch.rand(B, Nn, device=device) > 0.5)
            rdw, rdb = torch.empty(Kn, Nn, device=device), torch.empty(Nn, device=device)
            rws = torch.empty(max(int(lib.recalgo_dense_bwd_weights_workspace_bytes(B, Kn, Nn)), 16), dtype=torch.uint8, device=device)
            cx0, cw, cb, cg, cdx0, cws, cL = cross_ops
            dxr = torch.empty_like(xd)
            wsr = ops._wgrad_workspace(device, B, Kd, Nd, dwd)
            add(f"dense_bwd_rider(dense_bwd {Kd}->{Nd} + riders: dense_wgrad {Kn}->{Nn}, cross_bwd)",
                lambda: lib.recalgo_dense_bwd_rider(p(xd), Kd, p(gd), Nd, None, p(wd), B, Kd, Nd, None, 0, 0.0, p(dxr), Kd, p(dwd), p(dbd),
                                                    p(wsr), 1, None, None, None, None, p(xd), Kd, p(rx), Kn, p(rg), Nn, Kn, Nn, p(rdw),
                                                    p(rdb), p(rws), p(cx0), d, p(cw), p(cb), p(cg), d, d, cL, p(cdx0), p(cws), st),
                (B * (3 * Kd + 2 * Nd) + 2 * Kd * Nd) * 4 + (B * (Kn + Nn) + Kn * Nn) * 4 + B * 3 * d * 4, 2.0 * fl + 2.0 * B * Kn * Nn)
            res[-1]["prof"] = ["dense_bwd_rider_kernel", 0]
            res[-1]["grad_zero_fraction"] = 0.5
            keep.append((rx, rg, rdw, rdb, rws, dxr))
        res[-1]["grad_zero_fraction"] = 0.5
        res[-1]["mask_mode"] = "gradient arrives pre-masked (no y mask staged)" + ("; dx masked with the layer's input" if pm is not None else "")
        ops.dense_bwd(xd, gd, None, wd, dwd, dbd, defer=True, premask=pm)          # leaves this layer's split slabs + its pending entry
        keep.append((xd, gd, yd, wd, dwd, dbd))
        slab_bytes += int(lib.recalgo_dense_bwd_weights_workspace_bytes(B, Kd, Nd)) + (Kd * Nd + Nd) * 4
    # the step's deferred-sum launch: fixed-order sums of the three layers' split slabs (in the step it also sums the
    # loss tail's and CrossNet's partial rows and advances the optimizer's step counter)
    pending = list(ops._dense_pending)

    def sums_once():
        ops._dense_pending[:] = pending
        ops.flush_dense_splits()
    if pending:
        add(f"deferred_sums({len(pending)} layers' weight-gradient slabs)", sums_once, slab_bytes)
    # The optimizer launch (recalgo_adam_tf1_step: TF1 dense Adam over the flat dense buffer + over the arena's live-row
    # list; state as left by the timed steps; lr = 0 so that the repeated launches do not move the weights).  Rows no
    # gradient has ever reached have g = m = v = 0, for which the dense update is the identity.
    # Algorithmic bytes: 28 B per parameter of a live row (g, m, v, p in; p, m, v out) + 4 B per list entry + 28 B per
    # dense parameter.
    arenas = [a for a in store.arenas.values() if a.weight is not None and sp.plan_of(a) is None]
    n_dense = 0 if store.flat is None else store.flat.numel()
    live_bytes = sum(int(a.live_state()[2].item()) * (a.K * 28 + 4) for a in arenas)
    add("adam_tf1_step(dense variables" + (" + live-list arenas)" if arenas else ")"),
        lambda: ops.adam_tf1_step_(store.flat, store.flat_grad, store.flat_m, store.flat_v, arenas,
                                   store.opt_state["step"], None, 0.0, lazy=args.lazy_adam),
        live_bytes + n_dense * 28)
    res[-1]["live_fraction"] = round(live_fraction(est), 4)
    return res


def dense_layer_table(args, d):
    """The widths of the context MLP THIS model runs (d = fields x emb), and what stands in front of it:
    -> ([d_in, 512, 256, 128], None | "pnn" | "library").  DIN's fcn input is the 25 profile fields + target + attention
    output (din.py:221); PNN's MLP follows the 1024-unit product layer (pnn.py:133-193); FiBiNET's first layer reads the
    flattened [300 pairs x 32] interaction tensor (fibinet.py:177-199) and runs on hipBLASLt; NFM's MLP reads the K-wide
    bi-interaction vector; FwFM / AFM / FFM have no MLP (an empty table)."""
    F, K = args.fields, args.emb
    if args.model == "pnn":
        return [1024, 512, 256, 128], "pnn"
    if args.model == "fibinet":
        return [(F - 1) * (F - 2) // 2 * 2 * K, 512, 256, 128], "library"
    if args.model == "din":
        return [(F - 1) * K + 2 * K, 512, 256, 128], None
    if args.model == "nfm":
        return [K, 512, 256, 128], None
    if args.model in ("fwfm", "afm", "ffm"):
        return [], None
    return [d, 512, 256, 128], None


def cpu_quota():
    """CPUs the container may USE (cgroup CFS quota / period), as opposed to the hardware threads it can SEE: the GPU boxes of this
    pool show 256 hardware threads under a quota of 16 CPUs (`cpu.max` = 1600000 100000) — more runnable threads than that are
    throttled, which is what "the reader anti-scales beyond 32 threads" and "one oracle step takes 56 s at 256 threads" were.
    None: no quota."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except (OSError, ValueError):
        return None


def composite_roofline(ks, ms_per_step):
    """The whole step against its kernels' rooflines: sum over the step's hand-written kernels of t_min = max(algorithmic
    bytes / HBM peak, algorithmic FLOP / fp32 MFMA peak), divided by the measured step time (launch gaps, latency-bound
    kernels and everything else in the denominator).  A FUSED launch (DCN's tail, its rider launch) counts with the max over
    its summed bytes and FLOPs — less than the sum of the t_min of the launches it replaced; `frac_unfused` is the same ratio
    with those launches' own t_min instead (the accounting of the lines before the fusion: comparable across rounds)."""
    tm = lambda k: max(k["alg_bytes"] / (HBM_PEAK_GBS * 1e9), k.get("alg_flops", 0.0) / (FP32_PEAK_TFLOPS * 1e12))
    t_min = sum(tm(k) for k in ks if not k.get("part_of"))
    out = {"t_min_us": round(t_min * 1e6, 2), "ms_per_step": ms_per_step, "frac": round(t_min / (ms_per_step * 1e-3), 4),
           "kernels_counted": sum(1 for k in ks if not k.get("part_of"))}
    fused = {"tail_dense_head", "dense_bwd_rider"}
    if any(k.get("part_of") in fused for k in ks):
        t_un = sum(tm(k) for k in ks if (not k.get("part_of") and not any(k["kernel"].startswith(f + "(") for f in fused))
                   or (k.get("part_of") in fused and not k["kernel"].startswith("dense_wgrad(")))
        out["t_min_unfused_us"] = round(t_un * 1e6, 2)
        out["frac_unfused"] = round(t_un / (ms_per_step * 1e-3), 4)
    return out


def in_step_table(model: str):
    """The captured step as rocprofv3 saw it: every kernel a step dispatches, calls per step and average duration, from the
    newest committed `profiles/r*_<model>_kernel_stats.md` (scripts/rocpd_stats.py over `rocprofv3 --kernel-trace --stats` of
    this same command) — the ATen launches that are still in the step included.  None when no profile is committed."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{model}_kernel_stats.md")))
    if not files:
        return None
    rows, shapes, in_shapes = [], [], False
    with open(files[-1]) as f:
        for line in f:
            if line.startswith("per launch shape"):
                in_shapes = True
                continue
            if in_shapes:
                m = re.match(r"\| `(.*?)` \| (\d+) \| (\d+) \| (\d+) \|", line)      # kernel | grid | calls | avg_ns
                if m is not None:
                    shapes.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))))
                continue
            m = re.match(r"\| `(.*?)` \| (\d+) \| (\d+) \| (\d+) \|", line)
            if m is not None:
                rows.append((m.group(1), int(m.group(2)), int(m.group(4))))
    steps = next((c for n, c, _ in rows if "adam_tf1_step_kernel" in n), 0)
    if not steps:
        return None

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        return n.split("(")[0][:80]
    tab = [{"kernel": short(n), "calls_per_step": round(c / steps, 2), "avg_us": round(a / 1e3, 2)} for n, c, a in rows if c >= 0.9 * steps]
    return {"source": os.path.relpath(files[-1], ROOT), "steps_profiled": steps,
            "dispatches_per_step": round(sum(t["calls_per_step"] for t in tab), 1),
            "kernel_us_per_step": round(sum(t["calls_per_step"] * t["avg_us"] for t in tab), 1), "kernels": tab,
            "_shapes": [(n, g, c, a) for n, g, c, a in shapes if c >= 0.9 * steps] +
                       # (kernels the per-shape section does not list — it covers the dense / CIN / sparse families — have ONE launch
                       #  shape when they run once per step: their row of the main table is that shape)
                       [(n, 0, c, a) for n, c, a in rows if 0.9 * steps <= c <= 1.1 * steps and not any(n[:60] == sn[:60] for sn, _, _, _ in shapes)]}


def attach_in_step(ks, ist):
    """Rows of the live per-kernel table that name their kernel in the committed rocprofv3 table (`prof` = [substring of the
    kernel's name, rank among that kernel's launch shapes by average duration]) get the in-step average beside the isolated
    HIP-event one, and the roofline fraction at that duration: `in_step_avg_us`, `in_step_frac`."""
    if not ist:
        return
    for k in ks:
        pr = k.pop("prof", None)
        if not pr:
            continue
        # a launch shape two layers share (calls = 2 x steps) carries the average of both: ambiguous, no in-step figure
        steps = ist["steps_profiled"]
        cand = []
        for a, c in sorted(((a, c) for n, g, c, a in ist["_shapes"] if pr[0] in n), reverse=True):
            mult = max(1, int(round(c / steps)))
            cand += [(a, mult)] * mult
        if len(cand) > pr[1] and cand[pr[1]][1] == 1:
            us = cand[pr[1]][0] / 1e3
            t_min = max(k["alg_bytes"] / (HBM_PEAK_GBS * 1e9), k.get("alg_flops", 0.0) / (FP32_PEAK_TFLOPS * 1e12))
            k["in_step_avg_us"] = round(us, 2)
            k["in_step_frac"] = round(t_min / (us * 1e-6), 4)


def dominant_kernel(ks):
    """The step's largest single kernel: by its in-step duration in the committed rocprofv3 table where the row names one,
    else by the live isolated measurement."""
    return max((k for k in ks if k.get("launches", 1) == 1), key=lambda k: k.get("in_step_avg_us", k["avg_us"]))


def live_fraction(est) -> float:
    """Fraction of the embedding parameters whose row a gradient has reached (= what the optimizer walks every step)."""
    from recalgorithm_amd import sparse as sp
    tot = live = 0
    for a in est.store.arenas.values():
        if a.weight is None:
            continue
        tot += a.weight.numel()
        pl = sp.plan_of(a)
        if pl is not None:          # owner path: a row is live once it carries Adam state
            n_live = int((pl.last_step != 0).sum()) if pl.last_step is not None else int(((a.m != 0) | (a.v != 0)).any(dim=1).sum())
        else:
            n_live = int(a.live_state()[2].item())
        live += n_live * a.K
    return live / max(tot, 1)


def optimizer_state_sweep(args, r, device, rank, world):
    """The headline rotates `--data-batches` fixed batches, so its live-row set saturates.  Real training draws fresh
    batches: the rows TF1's dense Adam has to walk only grow.  This sweep continues the SAME run with fresh batches (one
    per step, never repeated; generated and moved to HBM outside the timed chunks) and reports the step time against the
    live fraction, then forces every row live (the dense pass a long run converges to)."""
    from recalgorithm_amd.io import synth
    est, spec, graphed = r["est"], r["spec"], r["graphed"]
    pts = [{"phase": f"headline ({args.data_batches} rotating batches)", "live_fraction": round(live_fraction(est), 4),
            "ms_per_step": round(r["dt"] / args.steps * 1e3, 4)}]
    run = (lambda b: graphed(*b)) if graphed is not None else (lambda b: est.train_step(*b))

    def timed(batches):
        """Median of the per-step device times of the chunk (a HIP event after every step): a host-side hiccup while the
        chunk is fed — an allocator or GC pause leaves the device idle — then does not pass for step time."""
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(batches) + 1)]
        evs[0].record()
        for i, bt in enumerate(batches):
            run(bt)
            evs[i + 1].record()
        evs[-1].synchronize()
        per = sorted(x.elapsed_time(y) for x, y in zip(evs, evs[1:]))
        return per[len(per) // 2]
    base = 100_000 + rank
    on_device = not (spec.with_history or spec.with_tags or spec.with_dense or max(spec.vocabs) > (1 << 24))
    total = args.sweep_batches if on_device else min(args.sweep_batches, 192)   # (host generator: ~15 batches/s)
    chunk = max(32, -(-total // 12))

    def fresh_batches(c0, n):
        if on_device:
            return synth.device_fresh_batches(spec, args.batch, device, n, seed=base + 7919 * world * c0)
        return [synth.device_features(spec, args.batch, device, batch_index=base + world * (c0 + i))[:2] for i in range(n)]
    for c0 in range(0, total, chunk):
        n = min(chunk, total - c0)
        fresh = fresh_batches(c0, n)
        ms = timed(fresh)
        pts.append({"phase": f"fresh batches {c0}..{c0 + n - 1} (never repeated)", "live_fraction": round(live_fraction(est), 4),
                    "ms_per_step": round(ms, 4)})
        del fresh
    from recalgorithm_amd import sparse as sp
    for a in est.store.arenas.values():
        if a.weight is None:
            continue
        pl = sp.plan_of(a)
        if pl is not None and pl.last_step is not None:
            # deferred-exact Adam: every row carries optimizer state (as after a long run) and is current — from here on
            # the sweep and the catch-ups replay real updates for all of them
            sp.sync(a, est.store.opt_state["step"], 0)
            never = pl.last_step == 0
            a.m[never] = 1e-3
            a.v[never] = 1e-6
            pl.last_step.copy_(est.store.opt_state["step"].to(torch.int32).expand_as(pl.last_step))
        elif pl is None and a.tracks_live_rows:
            a.force_all_live()
    fresh = fresh_batches(total, 24)
    timed(fresh[:4])
    pts.append({"phase": "every row forced live (TF1 dense Adam semantics over the whole table)", "live_fraction": round(live_fraction(est), 4),
                "ms_per_step": round(timed(fresh[4:]), 4)})
    return pts


def host_fed(args):
    """`host_fed`: the reference's own entry path (TFRecord file of string-keyed Examples + vocabulary files -> train_input_fn
    -> captured step) on a bounded synthetic file, as its own process: examples/s end to end, the reader
    alone, the host it ran on.  Never the headline `value` (whose inputs are resident in HBM)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_tfrecord.py"), "--examples", "65536", "--epochs", "80",
           "--batch", str(args.batch), "--fields", str(args.fields), "--max-vocab", str(args.max_vocab)]
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=240)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": "examples/s", "ms_per_step": d["ms_per_step"], "steps": d["steps"], "host": d["host"],
                "sample": f"DCN, {d['examples']} examples x {d['epochs']} epochs, shuffle buffer {d['shuffle_buffer']}, batch {args.batch}; "
                          "TFRecord bytes (26 string features + label per Example) -> native decode + vocabulary lookup -> one "
                          "staged host-to-device copy -> hipGraph replay"}
    except Exception as e:          # never take the headline line down
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_baseline(args, seconds):
    """The unfused op-for-op oracle (oracle/ref_ops.py, torch-CPU fp32) doing the same training
    step on a bounded sample of the workload.  Runs in a child process under a hard timeout so
    that a slow or over-subscribed host can never take the bench line down with it."""
    import subprocess
    if args.model == "ffm":
        return {"value": None, "unit": "examples/s", "cores": 0, "kind": "port",
                "sample": "not sampled: ONE oracle step (TF1 dense Adam over the 72.6 M rows of the 26 x 25 sub-tables, bag "
                          "walks in Python) takes > 60 s on the host — outside the bounded CPU sample of this bench"}
    def child(also_threads, secs, limit=None):
        cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--model", args.model, "--batch", str(args.batch),
               "--fields", str(args.fields), "--emb", str(args.emb), "--max-vocab", str(args.max_vocab),
               "--seconds", str(secs)] + (["--also-threads", str(also_threads)] if also_threads else [])
        limit = limit or max(150.0, 10 * secs)
        note, text = None, ""
        try:
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=limit)
            text = r.stdout or ""
            if r.returncode != 0:
                note = f"child failed rc={r.returncode}: {r.stderr.strip()[-300:]}"
        except subprocess.TimeoutExpired as e:
            text = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            note = f"child cut off after {limit:.0f} s (the samples it had finished by then are reported)"
        line = [l for l in text.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[-1])          # the child prints its result after every completed sample
            if note:
                d["note"] = note
            return d
        return {"value": None, "unit": "examples/s", "cores": 0, "kind": "port", "sample": note or "child printed nothing"}
    # ONE child (one import / build / warm-up): <= 32 threads (where the per-op work of one 4096-example batch stops scaling: the
    # better number on every box measured so far), then half of the host's hardware threads and ALL of them (SURVEY.md 8d's
    # definition); `value` / `cores` are those of the fastest sample, the others are `other_samples`
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    a = child(avail if avail > 32 else None, seconds)
    a["host_hardware_threads"] = avail
    a["host_cpu_quota"] = cpu_quota()         # (a cgroup quota below the thread count: the sample ran on that many CPUs' worth of time)
    others = a.get("other_samples") or []
    best = max(others, key=lambda o: o.get("value") or 0, default=None)
    if best and (best.get("value") or 0) > (a.get("value") or 0):
        a = dict(a)
        rest = [o for o in others if o is not best] + [{"cores": a["cores"], "value": a["value"], "unit": "examples/s",
                                                        "note": "the <= 32-thread sample: " + a["sample"][:120]}]
        a["value"], a["cores"], a["sample"] = best["value"], best["cores"], best["note"] + "; " + a["sample"]
        a["other_samples"] = rest
    if a.get("other_samples"):
        a["other_sample"] = a["other_samples"][-1]            # (the all-cores sample, under the key earlier rounds used)
    return a


def timed_run(args, device, rank, world, dist, capacity_factor):
    """Build the model on this rank, (N > 1) shard it, capture the step, warm up, time `steps` steps
    between barriers; returns the max-over-ranks wall time."""
    if args.tunable:
        import torch.cuda.tunable as tunable
        tunable.tuning_enable(True)
    shard = None
    if world > 1:
        # row-shard the embedding arenas over the ranks, BEFORE the build so that every rank only ever holds its own
        # rows (de-duplicated, fixed-capacity id / row all_to_all over RCCL: static shapes, no host sync -> the
        # N-GPU step is still one hipGraph), all-reduce the flat dense gradient, back-propagate loss / N
        from recalgorithm_amd.parallel import attach_data_parallel
        shard = lambda e: attach_data_parallel(e, dist, capacity_factor=capacity_factor)
    est, spec, feats, labels, workload = build_estimator(args, device, rank, world, before_build=shard)
    from recalgorithm_amd.estimator import GraphedTrainStep
    from recalgorithm_amd.io import synth
    # distinct synthetic batches, all resident in HBM before the timed region; step i consumes
    # batch i % n_batches (a device-to-device copy into the graph's static input buffers, inside
    # the timed region) so that the embedding rows touched differ from step to step
    batches = [(feats, labels)] + [
        synth.device_features(spec, args.batch, device, batch_index=rank + world * (1 + i))[:2]
        for i in range(args.data_batches - 1)]
    launch = "eager"
    graphed = None
    if not args.no_graph:
        try:
            graphed = GraphedTrainStep(est.train_step, feats, labels, warmup=3)
            launch = "hipGraph replay"
        except Exception as e:      # e.g. a collective that cannot be captured: run the same step eagerly
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager launches",
                  file=sys.stderr, flush=True)
            graphed = None
            torch.cuda.synchronize()
        if dist is not None:        # all ranks launch the same way: eager everywhere if any capture failed
            ok = torch.tensor([1.0 if graphed is not None else 0.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok) == 0.0:
                if not args.allow_eager:
                    # a scaling line measured on a silently degraded launch path would not be the design's number
                    raise SystemExit(f"[bench] rank {rank}: the {world}-rank step could not be captured into a hipGraph on every rank "
                                     "(see the message above); re-run with --allow-eager to time eager launches instead")
                graphed, launch = None, "eager"
    if graphed is None:
        from recalgorithm_amd.estimator import HOUSEKEEPING_EVERY

        def step(i):
            out = est.train_step(*batches[i % len(batches)])
            if i % HOUSEKEEPING_EVERY == HOUSEKEEPING_EVERY - 1:
                est.store.housekeeping()
            return out
        for i in range(max(args.warmup, 1)):
            loss = step(i)
    else:
        step = lambda i: graphed(*batches[i % len(batches)])
        for i in range(max(args.warmup - 3, 0)):
            step(i)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.tunable:
        tunable.tuning_enable(False)               # selections are frozen before the timed region
    barrier()
    marks = []                                      # a HIP event every 25 steps (no sync): per-chunk step times
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % 25 == 0:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((i, ev))
        loss = step(i)
    barrier()
    dt = time.perf_counter() - t0
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    end.synchronize()
    marks.append((args.steps, end))
    chunk_ms = [round(a[1].elapsed_time(b[1]) / max(b[0] - a[0], 1), 4) for a, b in zip(marks, marks[1:])]
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    overflow = False
    comm = None
    if world > 1:
        from recalgorithm_amd.parallel import exchange_overflowed
        ovf = torch.tensor([1.0 if exchange_overflowed(est) else 0.0], device=device)
        dist.all_reduce(ovf)
        overflow = float(ovf) > 0
        # what the line says about the communication that ran: the number of ranks that took part in a collective on the
        # benchmark's backend, and this rank's per-step exchange volume (static plan: fixed-capacity buckets, so the bytes
        # are a property of the configuration, not of the batch)
        one = torch.ones(1, device=device)
        dist.all_reduce(one)
        rows_b = ids_b = link_ids = link_rows = 0
        for ar in est.store.arenas.values():
            sd = getattr(ar, "sharding", None)
            if sd is None or sd.capacity_factor is None:
                continue
            n_req = args.batch * len(ar.tables)                        # one request per example and table of the arena
            cap = int(-(-n_req * sd.capacity_factor // world))
            ids_b += world * cap * 8                                   # local row numbers to the owners
            rows_b += 2 * world * cap * ar.K * 4                       # rows back, gradient rows forth
            link_ids += cap * 8                                        # ... of which ONE peer's bucket crosses ONE xGMI link
            link_rows += 2 * cap * ar.K * 4
        dense_b = 0 if est.store.flat_grad is None else est.store.flat_grad.numel() * 4
        comm = {"backend": "gloo (host-staged bring-up)" if os.environ.get("RECALGO_DIST_BACKEND") == "gloo_staged" else dist.get_backend(),
                "ranks_in_all_reduce": int(float(one)), "world_size": dist.get_world_size(),
                "per_rank_bytes_per_step": {"all_to_all_ids": int(ids_b), "all_to_all_rows_and_grads": int(rows_b),
                                            "all_reduce_dense_grads": int(dense_b)},
                # what ONE point-to-point xGMI link (~153 GB/s per direction) carries per step, for a one-glance sanity check of the
                # first real N-GPU run: an all_to_all is link-parallel on the full mesh (one peer bucket per link); a ring
                # all-reduce moves 2 (N - 1) / N of the buffer over every link of the ring
                "per_link_bytes_per_step": {"all_to_all_ids": int(link_ids), "all_to_all_rows_and_grads": int(link_rows),
                                            "all_reduce_dense_grads_ring": int(2 * (world - 1) * dense_b // world)},
                "per_link_us_at_153GBs": round((link_ids + link_rows + 2 * (world - 1) * dense_b / world) / 153e9 * 1e6, 2),
                "note": "static exchange plan: every bucket is sent at its fixed capacity (unused slots carry id -1 / zero rows)"}
    return {"est": est, "spec": spec, "feats": feats, "workload": workload, "dt": dt, "loss": float(loss), "graphed": graphed,
            "launch": launch, "overflow": overflow, "chunk_ms": chunk_ms, "step_fn": step, "comm": comm}


def extra_model(a, name, steps, device):
    """One more model of the target list in the same process: build, capture, warm up, time `steps` steps, per-kernel
    table; returns the compact entry of the bench line's `models` list."""
    a.model, a.steps, a.warmup, a.sweep_batches = name, steps, 10, 0
    r = timed_run(a, device, 0, 1, None, a.capacity_factor)
    e = {"model": name, "workload": r["workload"], "examples_per_s": round(a.batch * steps / r["dt"], 1),
         "ms_per_step": round(r["dt"] / steps * 1e3, 4), "steps": steps, "launch": r["launch"], "final_loss": round(r["loss"], 6),
         "live_fraction": round(live_fraction(r["est"]), 4)}
    cs = sorted(r["chunk_ms"])
    if cs:
        e["ms_per_step_p10_p50_p90"] = [cs[min(len(cs) - 1, int(q * len(cs)))] for q in (0.1, 0.5, 0.9)]
    if not a.no_kernel_timing:
        ks = kernel_rooflines(a, r["est"], r["feats"], device)
        ist = in_step_table(name)
        attach_in_step(ks, ist)
        dom = dominant_kernel(ks)
        e["roofline"] = {"kernel": dom["kernel"], "bound": dom["bound"], "frac": dom["frac"], "avg_us": dom["avg_us"],
                         "achieved": dom.get("achieved_TFLOPs", dom["achieved_GBs"]), "unit": "TFLOP/s" if dom["bound"] == "mfma" else "GB/s"}
        for key in ("in_step_avg_us", "in_step_frac"):
            if key in dom:
                e["roofline"][key] = dom[key]
        if ist:
            e["roofline"]["in_step_source"] = ist["source"]
        e["kernels"] = [{kk: k[kk] for kk in ("kernel", "avg_us", "bound", "frac", "in_step_avg_us", "in_step_frac", "part_of", "launches") if kk in k}
                        for k in ks]
        e["composite_roofline"] = composite_roofline(ks, e["ms_per_step"])
    return e


def main():
    args = parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)               # (does not return: the launcher's rank 0 prints the line)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line's n_gpus must be the number of ranks that ran")
    # RECALGO_DIST_BACKEND=gloo_staged: bring-up aid for boxes with fewer GPUs than ranks (ranks share
    # devices, collectives bounce through host memory; see parallel.HostStagedCollectives) — the
    # numbers it prints are not benchmark results.  Default: one rank per GPU over RCCL.
    staged = os.environ.get("RECALGO_DIST_BACKEND", "nccl") == "gloo_staged"
    if staged:
        local_rank %= torch.cuda.device_count()
        args.no_graph = True          # host-staged collectives synchronise: nothing to capture
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if staged:
            from recalgorithm_amd.parallel import HostStagedCollectives
            dist.init_process_group("gloo")
            dist = HostStagedCollectives(dist)
        else:
            dist.init_process_group("nccl", device_id=device)

    if args.tunable:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(50)        # ms per candidate
        tunable.set_max_tuning_iterations(20)
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"recalgo_tunableop_{rank}.csv"))
    # N > 1: the id/row exchange uses fixed-capacity buckets (capacity_factor x the mean bucket, so
    # that the step has static shapes and is one hipGraph).  A bucket that overflows invalidates
    # the run: it is repeated with the capacity doubled, up to `world` x the mean, at which every
    # bucket holds a whole batch and cannot overflow.
    cf = args.capacity_factor
    while True:
        r = timed_run(args, device, rank, world, dist, cf)
        if not r["overflow"] or world == 1:
            break
        if cf >= world:
            raise SystemExit("bench.py: exchange bucket overflow at full capacity (cannot happen)")
        if rank == 0:
            print(f"[bench] exchange bucket overflow at capacity factor {cf}: repeating with {min(2 * cf, world)}",
                  file=sys.stderr, flush=True)
        cf = min(2 * cf, float(world))
        del r
        torch.cuda.empty_cache()
    est, spec, feats, workload, dt, loss_v, launch = (r[k] for k in ("est", "spec", "feats", "workload", "dt", "loss", "launch"))

    out = {
        "metric": "CTR examples/sec (train step fwd+bwd+TF1-Adam), batch 4096/GPU, 26 fields x emb16",
        "value": round(world * args.batch * args.steps / dt, 1),
        "unit": "examples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "data_batches": args.data_batches, "global_batch": world * args.batch, "fields": args.fields,
                   "emb_dim": args.emb, "embedding_rows": int(sum(spec.vocabs)),
                   "optimizer": ("LazyAdam on the embedding tables (DEVIATION from the reference's tf.train.AdamOptimizer: rows without a "
                                 "gradient in a step keep weights and moments), Adam on the dense variables" if args.lazy_adam else
                                 "TF1 Adam, dense semantics over all tables, evaluated lazily but EXACTLY (csrc/sparse.hip: a row's g = 0 "
                                 "updates are replayed bit-identically when the row is next read, swept or flushed), fused with the "
                                 "row-gradient scatter"),
                   "launch": launch,
                   "dense_layers": ("hand-written fp32 MFMA (csrc/dense.hip + tile_v2.h); layers wider than 4096 inputs on hipBLASLt, "
                                    + ("TunableOp" if args.tunable else "default selection")),
                   "parallelism": (f"dp{world} + embedding rows sharded r % {world} (RCCL all_to_all), dense grads all-reduced"
                                   + (" [gloo_staged bring-up mode: NOT a benchmark]" if staged else "")
                                   if world > 1 else "single")},
        "final_loss": round(loss_v, 6),
        "ms_per_step_by_chunk_of_25": r["chunk_ms"],
    }
    if r.get("comm"):
        out["rccl_ranks"] = r["comm"]["ranks_in_all_reduce"]
        out["communication"] = r["comm"]
    if rank == 0:
        out["box"] = box_sanity(device)
        out["box"]["host_cores"] = os.cpu_count()
        out["box"]["host_cpu_quota"] = cpu_quota()       # CPUs the cgroup lets the process USE (None: no quota); host_cores = hardware threads visible
    cs = sorted(r["chunk_ms"])
    if cs:          # spread of the per-chunk step times (median and p10 / p90)
        pick = lambda q: cs[min(len(cs) - 1, int(q * len(cs)))]
        out["ms_per_step_p10_p50_p90"] = [pick(0.1), pick(0.5), pick(0.9)]
    ks = None
    if rank == 0 and not args.no_kernel_timing:
        # per-kernel table in the HEADLINE's state (before the optimizer-state sweep grows the live set); the repeated
        # lr = 0 optimizer launches decay the Adam moments, so those are put back afterwards
        snap = [(a, a.m.clone(), a.v.clone()) for a in est.store.arenas.values() if a.weight is not None]
        fm, fv = (None, None) if est.store.flat_m is None else (est.store.flat_m.clone(), est.store.flat_v.clone())
        try:
            ks = kernel_rooflines(args, est, feats, device)
        except Exception as e:          # the per-kernel table must never take the headline line down with it
            print(f"[bench] per-kernel timing failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        for a, m_, v_ in snap:
            a.m.copy_(m_); a.v.copy_(v_)
        if fm is not None:
            est.store.flat_m.copy_(fm); est.store.flat_v.copy_(fv)
        del snap, fm, fv
    if world == 1 and r["graphed"] is not None:
        # real percentiles for the headline state (the driver's --steps 20 is ONE chunk): 400 more steps on the same
        # rotating batches, a HIP event after every step
        try:
            step_fn = r["step_fn"]
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(401)]
            torch.cuda.synchronize()
            evs[0].record()
            for i in range(400):
                step_fn(i)
                evs[i + 1].record()
            evs[-1].synchronize()
            per = sorted(x.elapsed_time(y) for x, y in zip(evs, evs[1:]))
            out["headline_state_400_steps_ms_p10_p50_p90"] = [round(per[40], 4), round(per[200], 4), round(per[360], 4)]
        except Exception as e:
            print(f"[bench] percentile run failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
    sweep = None
    if args.sweep_batches > 0 and world == 1:
        try:
            sweep = optimizer_state_sweep(args, r, device, rank, world)
        except Exception as e:          # never take the headline line down
            print(f"[bench] optimizer-state sweep failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
    if sweep:
        out["optimizer_state_sweep"] = sweep
        last = sweep[-1]
        # what a long training run converges to: every row carries optimizer state
        out["steady_state"] = {"live_fraction": last["live_fraction"], "ms_per_step": last["ms_per_step"],
                               "examples_per_s": round(world * args.batch / (last["ms_per_step"] * 1e-3), 1),
                               "phase": last["phase"]}
    if rank == 0:
        if ks:
            ist = in_step_table(args.model)
            attach_in_step(ks, ist)
            for k in ks:
                k.pop("prof", None)
            dom = dominant_kernel(ks)                                                              # the dominant single KERNEL
            if dom["bound"] == "mfma":
                out["roofline"] = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved_TFLOPs"],
                                   "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"],
                                   "traffic": None, "avg_us": dom["avg_us"],
                                   "alg_flops_per_launch": dom["alg_flops"]}
            else:
                out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBs"],
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": None,
                                   "avg_us": dom["avg_us"], "alg_bytes_per_launch": dom["alg_bytes"]}
            # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this
            # round (FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per
            # guides/MI355X_MICROARCH.md: gfx950 counts a wide coalesced read at half its bytes)
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    t = json.load(f).get(args.model, {}).get(dom["kernel"])
                # only a measurement taken in (nearly) this run's state is comparable with this run's algorithmic bytes
                if t and abs(t.get("alg_bytes", 0) - dom["alg_bytes"]) <= 0.1 * dom["alg_bytes"]:
                    out["roofline"]["traffic"] = int(2 * t["fetch_size_kb"] * 1024 + t["write_size_kb"] * 1024)
                    out["roofline"]["traffic_source"] = t["source"]
            except (OSError, ValueError, KeyError):
                pass
            out["kernels"] = ks
            # the step as profiled (isolated HIP-event numbers above are per kernel in a loop of its own; this is the captured
            # step under rocprofv3: name for name what a step dispatches, ATen launches included)
            # (a COMMITTED profile of an earlier run of this command on another box — `source` says which file —, not a measurement
            # of this run: hence the key's name)
            for key in ("in_step_avg_us", "in_step_frac"):
                if key in dom:
                    out["roofline"][key] = dom[key]
            if ist:
                out["roofline"]["in_step_source"] = ist["source"]
                ist.pop("_shapes", None)
                out["committed_profile"] = ist
            out["composite_roofline"] = composite_roofline(ks, out["ms_per_step"])
            # whole-step view (BASELINE.json: "absolute and as fraction of HBM roofline"): the algorithmic
            # bytes of the step's HBM-bound hand-written kernels over the WHOLE step time, library GEMMs
            # and launch gaps included in the denominator
            hb = sum(k["alg_bytes"] for k in ks if k["bound"] == "hbm")
            out["step_hbm"] = {"alg_bytes_hot_path_kernels": int(hb), "ms_per_step": out["ms_per_step"],
                               "achieved_GBs": round(hb / (out["ms_per_step"] * 1e-3) / 1e9, 1),
                               "frac_of_peak": round(hb / (out["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if args.model == "dcn" and world == 1 and not args.no_extra_models and not args.big_table_rows:
            # the other three models BASELINE.json's target names, each its own build / capture / timed run in this process
            del est, r
            torch.cuda.empty_cache()
            out["models"] = [{"model": "dcn", "examples_per_s": out["value"], "ms_per_step": out["ms_per_step"],
                              "roofline": {k: out["roofline"][k] for k in ("kernel", "bound", "frac", "avg_us")} if "roofline" in out else None,
                              "composite_roofline": out.get("composite_roofline")}]
            import copy
            for name, steps in (("deepfm", 300), ("xdeepfm", 60), ("din", 200)):
                try:
                    out["models"].append(extra_model(copy.copy(args), name, steps, device))
                except Exception as e:      # never take the headline line down
                    out["models"].append({"model": name, "error": f"{type(e).__name__}: {e}"})
                torch.cuda.empty_cache()
        if not args.no_host_fed and world == 1 and args.model == "dcn" and not args.big_table_rows:
            out["host_fed"] = host_fed(args)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        (dist._d if staged else dist).destroy_process_group()


if __name__ == "__main__":
    main()
