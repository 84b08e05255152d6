"""oracle/gen_golden.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Generates tests/golden/*.npz by importing and EXECUTING the reference's own, unmodified
sources from /root/reference/algorithm against oracle/tf1_shim (a torch-CPU stand-in for the
TF-1.14 API they call; TensorFlow itself cannot be installed here).  Run in the authoring
container only (the GPU box has no /root/reference):

    python -m oracle.gen_golden          # rewrites tests/golden/

Two families of vectors (all float64, seeded):
  layer_*.npz  the reference's layer functions called directly:
               cross_layer, cin_layer, din_attention, prelu, dice, senet,
               bilinear_interaction_layer  -> inputs, variables (by TF name), outputs, and the
               gradients of sum(out * G) wrt inputs and variables
  model_*.npz  the reference's <model>_model_fn(features, labels, mode, params) in PREDICT and
               TRAIN mode on a 48-example WeChat-shaped batch built from the reference's own
               create_feature_columns() over a synthetic vocabulary directory: string features,
               variables before, predictions, loss, every gradient, variables after ONE run of
               the reference's train_op (TF1 Adam)

What this pins: the reference's composition (op order, axes, variable names and shapes,
quirks).  What it cannot pin: TF's kernels (oracle/__init__.py "Pinning status").
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/algorithm"
OUT = os.path.join(ROOT, "tests", "golden")

VOCABS = {  # file stem -> number of keys "<stem>_<i>"
    "userid": 37, "feedid": 53, "device": 2, "authorid": 29, "bgm_song_id": 23, "bgm_singer_id": 19,
    "manual_tag_id": 31,
}
DENSE = ["videoplayseconds", "u_read_comment_7d_sum", "u_like_7d_sum", "u_click_avatar_7d_sum",
         "u_forward_7d_sum", "u_comment_7d_sum", "u_follow_7d_sum", "u_favorite_7d_sum",
         "i_read_comment_7d_sum", "i_like_7d_sum", "i_click_avatar_7d_sum", "i_forward_7d_sum",
         "i_comment_7d_sum", "i_follow_7d_sum", "i_favorite_7d_sum", "c_user_author_read_comment_7d_sum"]


def _use_shim():
    shim = os.path.join(HERE, "tf1_shim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    import tensorflow as tf
    assert "tf1_shim" in tf.__file__, tf.__file__
    return tf


def _import_ref(model_dir: str, module: str):
    """Import /root/reference/algorithm/<model_dir>/<module>.py exactly as `python <module>.py`
    run from inside that directory would see it (cwd-relative sys.path hacks included)."""
    d = os.path.join(REF, model_dir)
    cwd = os.getcwd()
    os.chdir(d)
    sys.path.insert(0, d)
    try:
        for stale in ("utils", module):
            sys.modules.pop(stale, None)
        return importlib.import_module(module)
    finally:
        sys.path.remove(d)
        os.chdir(cwd)


def write_vocab_dir(path: str):
    os.makedirs(path, exist_ok=True)
    for stem, n in VOCABS.items():
        with open(os.path.join(path, stem + ".txt"), "w") as f:
            for i in range(n):
                f.write(f"{stem}_{i}\n")


def make_batch(B: int, seed: int):
    """WeChat-shaped string features (DataGenerator.py:403-443 field set): 6 single-valued ids
    ('' = missing -> OOV), the multi-valued tag list and the read-comment history (shares the
    feedid vocabulary), 16 dense floats, the label."""
    rng = np.random.default_rng(seed)
    feats = {}

    def draw(stem, vocab_stem=None, oov=0.08):
        vs = vocab_stem or stem
        k = int(rng.integers(0, VOCABS[vs]))
        r = rng.random()
        if r < oov / 2:
            return ""
        if r < oov:
            return f"{vs}_{VOCABS[vs] + 5}"          # a key that is not in the vocabulary file
        return f"{vs}_{k}"
    for key in ("userid", "feedid", "device", "authorid", "bgm_song_id", "bgm_singer_id"):
        feats[key] = [[draw(key)] for _ in range(B)]
    feats["manual_tag_list"] = [[draw("manual_tag_list", "manual_tag_id") for _ in range(int(rng.integers(0, 5)))]
                                for _ in range(B)]
    feats["his_read_comment_7d_seq"] = [[draw("his", "feedid") for _ in range(int(rng.integers(0, 9)))]
                                        for _ in range(B)]
    feats["his_read_comment_7d_seq"][0] = []                                  # length-0 history (din_attention.py:52)
    dense = np.log1p(rng.poisson(3.0, size=(B, len(DENSE)))).astype(np.float64)
    labels = (rng.random((B, 1)) < 0.3).astype(np.float64)
    return feats, dense, labels


def _ragged(rows):
    vals = [w for r in rows for w in r]
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    return np.array(vals if vals else [""], dtype="U32")[:len(vals)], offs


def _np(x):
    x = x.t if hasattr(x, "t") and isinstance(getattr(x, "t"), torch.Tensor) else x
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


# =============================================================================================
# layer-level vectors
# =============================================================================================
def layer_goldens(tf):
    out = {}
    gen = torch.Generator().manual_seed(2024)
    rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)

    def run(name, fn, inputs, scope=None, extra=None):
        """inputs: dict name -> torch tensor (float: differentiated).  fn(**T inputs) -> T."""
        tf.reset_default_graph(seed=zlib.crc32(name.encode()) % 10000)
        tin = {}
        for k, v in inputs.items():
            t = v.clone()
            if t.is_floating_point():
                t.requires_grad_(True)
            tin[k] = tf.T(t)
        if scope:
            with tf.variable_scope(scope):
                y = fn(**tin)
        else:
            y = fn(**tin)
        G = rnd(*y.t.shape)
        (y.t * G).sum().backward()
        d = {f"in/{k}": _np(v) for k, v in inputs.items()}
        d["out"] = _np(y)
        d["G"] = _np(G)
        for k, v in tin.items():
            if v.t.is_floating_point():
                d[f"grad_in/{k}"] = _np(v.t.grad if v.t.grad is not None else torch.zeros_like(v.t))
        for vn, var in tf.get_default_graph().vars.items():
            d[f"var/{vn}"] = _np(var)
            d[f"grad_var/{vn}"] = _np(var.t.grad if var.t.grad is not None else torch.zeros_like(var.t))
        for k, v in (extra or {}).items():
            d[f"meta/{k}"] = np.asarray(v)
        out[name] = d

    # DCN cross layer stack, as dcn.py:157-160 drives it
    cross = _import_ref("DCN", "cross_layer")

    def cross_stack(x0):
        xl = x0
        for i in range(3):
            xl = cross.cross_layer(x0=x0, xl=xl, index=i)
        return xl
    run("layer_cross_stack", cross_stack, {"x0": rnd(9, 24) * 0.5}, scope="cross_part")
    run("layer_cross_single", lambda x0, xl: cross.cross_layer(x0, xl, 7), {"x0": rnd(5, 12), "xl": rnd(5, 12)})

    # xDeepFM CIN, as xdeepfm.py:166-174 drives it (index starts at 1; maps arrive as strings)
    cin = _import_ref("xDeepFM", "cin_layer")

    def cin_stack(x0):
        xk, pools = x0, []
        for i, h in enumerate(["6", "5"]):
            xk = cin.cin_layer(x0, xk, h, i + 1)
            pools.append(tf.reduce_sum(xk, axis=-1))
        return tf.concat(pools, axis=-1)
    run("layer_cin_stack", cin_stack, {"x0": rnd(7, 5, 8) * 0.7}, scope="cin_part")
    run("layer_cin_single", lambda x0, xk: cin.cin_layer(x0, xk, 4, 3), {"x0": rnd(6, 5, 4), "xk": rnd(6, 3, 4)})

    # DIN attention, both branches, lengths incl. 0 and full
    att = _import_ref("DIN", "din_attention")
    lens = torch.tensor([0, 1, 6, 3, 6, 2, 5], dtype=torch.int64)
    keys = rnd(7, 6, 16) * 0.6
    for b, L in enumerate(lens.tolist()):
        keys[b, L:] = 0.0                                 # sequence_input_layer pads with exact zeros
    for sm in (False, True):
        run(f"layer_din_attention_{'softmax' if sm else 'default'}",
            lambda query, keys, keys_length, sm=sm: att.din_attention(query, keys, keys_length, is_softmax=sm),
            {"query": rnd(7, 16) * 0.6, "keys": keys, "keys_length": lens}, scope="attention_part",
            extra={"is_softmax": sm})

    act = _import_ref("DIN", "activations")

    def with_alpha(fn, vname):
        def f(x):
            y0 = fn(x, name=1)                             # creates the alpha variable (init 1.0)
            g = tf.get_default_graph()
            with torch.no_grad():
                g.vars[vname].t.copy_(torch.linspace(0.1, 0.9, x.t.shape[-1], dtype=torch.float64))
            return fn(x, name=1)
        return f
    run("layer_prelu", with_alpha(act.prelu, "prelu_alpha_1"), {"x": rnd(11, 6)})
    run("layer_dice", with_alpha(act.dice, "dice_alpha_1"), {"x": rnd(11, 6)})

    # FiBiNET
    se = _import_ref("FiBiNET", "senet")
    run("layer_senet", lambda input: se.senet(input, 8, 2), {"input": rnd(6, 7, 8)}, scope="senet_part")
    bi = _import_ref("FiBiNET", "bilinear_interaction_layer")
    for ty in ("all", "each", "interaction"):
        run(f"layer_bilinear_{ty}", lambda input, ty=ty: bi.bilinear_interaction_layer(input, 8, ty, "orginal"),
            {"input": rnd(5, 7, 8) * 0.8}, scope="bilinear_interaction_part", extra={"type": ty})
    return out


# =============================================================================================
# model-level vectors
# =============================================================================================
def model_goldens(tf, vocab_dir):
    out = {}
    B = 48
    sfeats, dense, labels = make_batch(B, seed=77)

    def features_for(cols):
        f = {}
        for c in cols:
            if c.key in sfeats:
                f[c.key] = sfeats[c.key]
            elif c.key in DENSE:
                f[c.key] = tf.T(torch.from_numpy(dense[:, DENSE.index(c.key)].reshape(-1, 1).copy()))
        return f

    def run(name, module, model_fn_name, make_params, flag_overrides):
        for k, v in flag_overrides.items():
            setattr(module.FLAGS, k, v)
        module.FLAGS.vocabulary_dir = vocab_dir
        params, all_cols = make_params(module)
        model_fn = getattr(module, model_fn_name)
        M = tf.estimator.ModeKeys
        d = {}
        # PREDICT on a fresh graph; variables are created here
        tf.reset_default_graph(seed=4242)
        feats = features_for(all_cols)
        spec = model_fn(feats, None, M.PREDICT, params)
        g = tf.get_default_graph()
        for vn, var in g.vars.items():
            d[f"var/{vn}"] = _np(var).copy()
        for k, v in spec.predictions.items():
            d[f"predict/{k}"] = _np(v)
        # TRAIN on the same variables (a second model_fn call == a new TF graph: naming restarts)
        g.uid.clear(); g.collections.clear(); g.scope.clear()
        lab = {"read_comment": tf.T(torch.from_numpy(labels.copy()))}
        spec = model_fn(feats, lab, M.TRAIN, params)
        d["train/loss"] = _np(spec.loss)
        for i, mk in enumerate(g.collections.get("__dropout_masks__", [])):       # training-mode dropout keep masks, call order
            d[f"aux/dropout_mask_{i}"] = mk.numpy().copy()
        grads = spec.train_op.run()
        for vn, gv in grads.items():
            d[f"grad/{vn}"] = _np(gv)
        for vn, var in g.vars.items():
            d[f"var_after/{vn}"] = _np(var).copy()
        # EVAL after the step
        g.uid.clear(); g.collections.clear(); g.scope.clear()
        spec = model_fn(feats, lab, M.EVAL, params)
        d["eval/loss"] = _np(spec.loss)
        d["eval/accuracy"] = _np(spec.eval_metric_ops["eval_accuracy"][0])
        d["eval/auc"] = _np(spec.eval_metric_ops["eval_auc"][0])
        for k, v in flag_overrides.items():
            d[f"flag/{k}"] = np.asarray(v)
        d["meta/learning_rate"] = np.asarray(params["learning_rate"])
        out[name] = d

    common = {"hidden_units": "16,8", "learning_rate": 0.005}

    def deepfm_params(m):
        first, second, label = m.create_feature_columns()
        return ({"first_order_feature_columns": first, "second_order_feature_columns": second,
                 "hidden_units": m.FLAGS.hidden_units.split(","), "learning_rate": m.FLAGS.learning_rate,
                 "dropout_rate": m.FLAGS.dropout_rate, "batch_norm": m.FLAGS.batch_norm}, first + second)
    run("model_deepfm", _import_ref("DeepFM", "deepfm"), "deepfm_model_fn", deepfm_params,
        dict(common, embedding_dim=8, dropout_rate=0.0, batch_norm=True))

    # the reference's DEFAULT training configuration has dropout_rate = 0.1 (deepfm.py:39): dense(relu) -> dropout -> BN;
    # the keep masks the run drew are part of the golden (aux/dropout_mask_<i>, call order)
    run("model_deepfm_dropout", _import_ref("DeepFM", "deepfm"), "deepfm_model_fn", deepfm_params,
        dict(common, embedding_dim=8, dropout_rate=0.1, batch_norm=True))

    def dcn_params(m):
        dense_c, cat, label = m.create_feature_columns()
        return ({"category_feature_columns": cat, "dense_feature_columns": dense_c,
                 "hidden_units": m.FLAGS.hidden_units.split(","), "num_cross_layer": m.FLAGS.num_cross_layer,
                 "learning_rate": m.FLAGS.learning_rate}, dense_c + cat)
    run("model_dcn", _import_ref("DCN", "dcn"), "dcn_model_fn", dcn_params, dict(common, num_cross_layer=3))

    def xdeepfm_params(m):
        dense_c, cat, label = m.create_feature_columns()
        return ({"category_feature_columns": cat, "dense_feature_columns": dense_c,
                 "hidden_units": m.FLAGS.hidden_units.split(","), "learning_rate": m.FLAGS.learning_rate,
                 "embedding_dim": m.FLAGS.embedding_dim,
                 "cin_layer_feature_maps": m.FLAGS.cin_layer_feature_maps.split(",")}, dense_c + cat)
    run("model_xdeepfm", _import_ref("xDeepFM", "xdeepfm"), "xdeepfm_model_fn", xdeepfm_params,
        dict(common, embedding_dim=8, cin_layer_feature_maps="6,5"))

    def din_params(m):
        dense_c, cat, tgt, seq, label = m.create_feature_columns()
        F = m.FLAGS
        return ({"dense_feature_columns": dense_c, "category_feature_columns": cat, "sequence_feature_columns": seq,
                 "target_feedid_feature_columns": tgt, "hidden_units": F.hidden_units.split(","),
                 "dropout_rate": F.dropout_rate, "batch_norm": F.batch_norm, "learning_rate": F.learning_rate,
                 "activation": F.activation, "mini_batch_aware_regularization": F.mini_batch_aware_regularization,
                 "l2_lambda": F.l2_lambda, "use_softmax": F.use_softmax}, dense_c + cat + tgt + seq)
    din = _import_ref("DIN", "din")
    run("model_din_dice", din, "din_model_fn", din_params,
        dict(common, dropout_rate=0.0, batch_norm=True, activation="dice", mini_batch_aware_regularization=True,
             l2_lambda=0.2, use_softmax=False))
    run("model_din_prelu_softmax", din, "din_model_fn", din_params,
        dict(common, dropout_rate=0.0, batch_norm=True, activation="prelu", mini_batch_aware_regularization=False,
             l2_lambda=0.2, use_softmax=True))

    run("model_din_dice_dropout", din, "din_model_fn", din_params,                         # din.py:41 default rate; BN -> dropout
        dict(common, dropout_rate=0.1, batch_norm=True, activation="dice", mini_batch_aware_regularization=True,
             l2_lambda=0.2, use_softmax=False))

    def fibinet_params(m):
        dense_c, cat, label = m.create_feature_columns()
        F = m.FLAGS
        return ({"category_feature_columns": cat, "dense_feature_columns": dense_c,
                 "hidden_units": F.hidden_units.split(","), "dropout_rate": F.dropout_rate,
                 "batch_norm": F.batch_norm, "learning_rate": F.learning_rate, "embedding_dim": F.embedding_dim,
                 "reduction_ratio": F.reduction_ratio, "bilinear_interaction_type": F.bilinear_interaction_type},
                dense_c + cat)
    fib = _import_ref("FiBiNET", "fibinet")
    for ty in ("all", "each", "interaction"):
        run(f"model_fibinet_{ty}", fib, "fibinet_model_fn", fibinet_params,
            dict(common, embedding_dim=8, dropout_rate=0.0, batch_norm=True, reduction_ratio=2,
                 bilinear_interaction_type=ty))

    run("model_fibinet_all_dropout", fib, "fibinet_model_fn", fibinet_params,                 # fibinet.py:42 default rate
        dict(common, embedding_dim=8, dropout_rate=0.1, batch_norm=True, reduction_ratio=2,
             bilinear_interaction_type="all"))

    def pnn_params(m):
        cat, label = m.create_feature_columns()
        F = m.FLAGS
        return ({"category_feature_columns": cat, "hidden_units": F.hidden_units.split(","),
                 "dropout_rate": F.dropout_rate, "batch_norm": F.batch_norm, "learning_rate": F.learning_rate,
                 "output_dimension": F.output_dimension, "product_method": F.product_method,
                 "weight_regularizer": F.weight_regularizer}, cat)
    pnn = _import_ref("PNN", "pnn")
    run("model_pnn_ipnn", pnn, "pnn_model_fn", pnn_params,
        dict(common, embedding_dim=8, dropout_rate=0.0, batch_norm=True, output_dimension=20,
             product_method="IPNN", weight_regularizer=0.0))
    run("model_pnn_opnn_reg", pnn, "pnn_model_fn", pnn_params,
        dict(common, embedding_dim=8, dropout_rate=0.0, batch_norm=True, output_dimension=20,
             product_method="OPNN", weight_regularizer=0.01))

    run("model_pnn_ipnn_dropout", pnn, "pnn_model_fn", pnn_params,                            # pnn.py:39 default rate
        dict(common, embedding_dim=8, dropout_rate=0.1, batch_norm=True, output_dimension=20,
             product_method="IPNN", weight_regularizer=0.0))
    run("model_pnn_ipnn_dropout_nobn", pnn, "pnn_model_fn", pnn_params,                       # dropout feeding a dense layer directly
        dict(common, embedding_dim=8, dropout_rate=0.25, batch_norm=False, output_dimension=20,
             product_method="IPNN", weight_regularizer=0.0))

    # §8f-3 sibling: FwFM (first-order dense over indicators + field-pair-weighted inner products, no MLP)
    def fwfm_params(m):
        first, second, label = m.create_feature_columns()
        return ({"first_order_feature_columns": first, "second_order_feature_columns": second,
                 "embedding_dim": m.FLAGS.embedding_dim, "learning_rate": m.FLAGS.learning_rate}, first + second)
    run("model_fwfm", _import_ref("FwFM", "fwfm"), "fwfm_model_fn", fwfm_params,
        dict(learning_rate=0.005, embedding_dim=8))

    # §8f-3 sibling: AFM (pair Hadamard products + attention net, afm.py:143-190)
    def afm_params(m):
        dense_c, cat, label = m.create_feature_columns()
        return ({"category_feature_columns": cat, "dense_feature_columns": dense_c,
                 "embedding_dim": m.FLAGS.embedding_dim, "attention_factor": m.FLAGS.attention_factor,
                 "learning_rate": m.FLAGS.learning_rate}, dense_c + cat)
    run("model_afm", _import_ref("AFM", "afm"), "afm_model_fn", afm_params,
        dict(learning_rate=0.005, embedding_dim=8, attention_factor=12))

    # §8f-3 sibling: FFM (field-aware pair inner products, ffm.py:118-163)
    def ffm_params(m):
        cols, label = m.create_feature_columns()
        return ({"one_hot_category_feature_columns": cols, "learning_rate": m.FLAGS.learning_rate,
                 "embedding_dim": m.FLAGS.embedding_dim,
                 "fields_vocabulary_size_tuple": [(c.categorical_column.name, int(c.variable_shape[-1])) for c in cols]}, cols)
    run("model_ffm", _import_ref("FFM", "ffm"), "ffm_model_fn", ffm_params, dict(learning_rate=0.005, embedding_dim=4))

    # §8f-3 sibling: NFM (bi-interaction pooling + MLP, nfm.py:143-182); its
    # dropout after the pooling is hard-coded (0.1, :170): the TRAIN golden carries the keep mask
    def nfm_params(m):
        dense_c, cat, label = m.create_feature_columns()
        return ({"category_feature_columns": cat, "dense_feature_columns": dense_c,
                 "hidden_units": m.FLAGS.hidden_units.split(","), "learning_rate": m.FLAGS.learning_rate,
                 "dropout_rate": m.FLAGS.dropout_rate, "batch_norm": m.FLAGS.batch_norm}, dense_c + cat)
    run("model_nfm", _import_ref("NFM", "nfm"), "nfm_model_fn", nfm_params,
        dict(common, embedding_dim=8, dropout_rate=0.0, batch_norm=True))
    run("model_nfm_dropout", _import_ref("NFM", "nfm"), "nfm_model_fn", nfm_params,          # nfm.py:41 default rate: dense -> BN -> dropout
        dict(common, embedding_dim=8, dropout_rate=0.1, batch_norm=True))

    # the shared batch
    batch = {"dense": dense, "labels": labels, "dense_names": np.array(DENSE)}
    for k, rows in sfeats.items():
        v, o = _ragged(rows)
        batch[f"str/{k}/values"], batch[f"str/{k}/offsets"] = v, o
    for stem, n in VOCABS.items():
        batch[f"vocab/{stem}"] = np.asarray(n)
    out["batch"] = batch
    return out


def main():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden.py needs /root/reference (authoring container only)")
    tf = _use_shim()
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as vd:
        vocab_dir = os.path.join(vd, "vocabulary") + "/"
        write_vocab_dir(vocab_dir)
        allg = {}
        allg.update(layer_goldens(tf))
        allg.update(model_goldens(tf, vocab_dir))
    only = [a for a in sys.argv[1:] if not a.startswith("-")]       # e.g. `gen_golden.py model_fwfm`: write these only
    for name, d in allg.items():
        if only and name not in only:
            continue
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(f"[golden] {name}.npz  ({len(d)} arrays)")


if __name__ == "__main__":
    main()
