"""-m gpu: the MI355X path (mirrored layer functions and model_fns -> C-ABI -> HIP kernels) against
the golden vectors produced by running the reference's OWN sources on oracle/tf1_shim
(oracle/gen_golden.py; float64).  Tolerance: north_star's 1e-5 relative (tests/util.py)."""
import re

import numpy as np
import pytest
import torch

from recalgorithm_amd.estimator import Estimator, ModeKeys, RunConfig
from recalgorithm_amd.variables import VariableStore, named_grads, use_store, variable_scope
from tests import golden_util as GU
from tests.util import assert_adam_update, assert_close

pytestmark = pytest.mark.gpu


def _layer(dev, name, fn, scope=None, var_names=None):
    """fn(inputs on device) -> out; run once to create the variables, load the golden values,
    run again, back-propagate the golden upstream gradient, compare everything."""
    d = GU.load(name)
    ins = {}
    for k, v in GU.section(d, "in/").items():
        t = torch.from_numpy(v.copy())
        t = t.float().to(dev).requires_grad_(True) if t.is_floating_point() else t.to(dev)
        ins[k] = t
    store = VariableStore(dev, seed=1)

    def call():
        store.begin_call()
        if scope:
            with variable_scope(scope):
                return fn(ins)
        return fn(ins)
    with use_store(store):
        with torch.no_grad():
            call()
        gv = GU.section(d, "var/")
        for vn, val in gv.items():
            if "dice_bn" in vn:
                continue            # Dice's never-updated BN statistics (0, 1) are constants of the kernel
            v = store.vars[vn]
            v.data.copy_(torch.from_numpy(val).float().reshape(v.data.shape))
        out = call()
    assert_close(out, torch.from_numpy(d["out"]), what=f"{name} out")
    out.backward(torch.from_numpy(d["G"]).float().to(dev))
    from recalgorithm_amd import nn as _nn
    _nn.apply_parked_grads()          # deferred sums (weight-gradient splits, CrossNet's partial rows): what the optimizer does
    for k, g in GU.section(d, "grad_in/").items():
        assert_close(ins[k].grad, torch.from_numpy(g), what=f"{name} d(in {k})", reduced=True)
    for vn, g in GU.section(d, "grad_var/").items():
        if "dice_bn" in vn:
            continue
        sib = vn.replace("/bias", "/kernel")
        gsec = GU.section(d, "grad_var/")
        # f3_att/bias = sum_t ds_t is exactly 0 under softmax: judged at its sibling kernel's scale
        floor = 1e-5 * float(np.abs(gsec[sib]).max()) if vn.endswith("/bias") and sib in gsec else 0.0
        assert_close(store.vars[vn].grad, torch.from_numpy(g), what=f"{name} d(var {vn})", reduced=True, floor=floor)


def test_cross_layer_golden(dev):
    from recalgorithm_amd.algorithm.DCN.cross_layer import cross_layer, cross_network

    def stack(i):                       # the reference's loop, layer by layer (dcn.py:157-160)
        xl = i["x0"]
        for l in range(3):
            xl = cross_layer(x0=i["x0"], xl=xl, index=l)
        return xl
    _layer(dev, "layer_cross_stack", stack, scope="cross_part")
    _layer(dev, "layer_cross_single", lambda i: cross_layer(i["x0"], i["xl"], 7))


def test_cross_network_fused_golden(dev):
    """The fused L-layer kernel against the same golden (variables wl_i / bl_i live in one block)."""
    from recalgorithm_amd.algorithm.DCN.cross_layer import cross_network
    _layer(dev, "layer_cross_stack", lambda i: cross_network(i["x0"], 3), scope="cross_part")


def test_cin_layer_golden(dev):
    from recalgorithm_amd.algorithm.xDeepFM.cin_layer import cin_layer, cin_network

    def stack(i):
        _, p_plus = cin_network(i["x0"], ["6", "5"])
        return p_plus
    _layer(dev, "layer_cin_stack", stack, scope="cin_part")
    _layer(dev, "layer_cin_single", lambda i: cin_layer(i["x0"], i["xk"], 4, 3))


@pytest.mark.parametrize("branch", ["default", "softmax"])
def test_din_attention_golden(dev, branch):
    from recalgorithm_amd.algorithm.DIN.din_attention import din_attention
    _layer(dev, f"layer_din_attention_{branch}",
           lambda i: din_attention(i["query"], i["keys"], i["keys_length"], is_softmax=(branch == "softmax")),
           scope="attention_part")


def test_activations_golden(dev):
    from recalgorithm_amd.algorithm.DIN.activations import dice, prelu
    _layer(dev, "layer_prelu", lambda i: prelu(i["x"], name=1))
    _layer(dev, "layer_dice", lambda i: dice(i["x"], name=1))


def test_fibinet_layers_golden(dev):
    from recalgorithm_amd.algorithm.FiBiNET.bilinear_interaction_layer import bilinear_interaction_layer
    from recalgorithm_amd.algorithm.FiBiNET.senet import senet
    _layer(dev, "layer_senet", lambda i: senet(i["input"], 8, 2), scope="senet_part")
    for ty in ("all", "each", "interaction"):
        _layer(dev, f"layer_bilinear_{ty}", lambda i, ty=ty: bilinear_interaction_layer(i["input"], 8, ty, "orginal"),
               scope="bilinear_interaction_part")
    with pytest.raises(ValueError):
        with use_store(VariableStore(dev)):
            bilinear_interaction_layer(torch.zeros(2, 5, 8, device=dev), 8, "bogus", "x")


@pytest.mark.parametrize("name", GU.MODELS)
def test_model_golden(dev, name, tmp_path):
    vocab_dir = GU.write_vocab_dir(str(tmp_path / "vocabulary"))
    model_fn, params, oracle_name = GU.mirror_setup(name, vocab_dir)
    d = GU.load(name)
    sfeats, labels = GU.string_batch()
    # the reference arithmetic's own fp32 rounding on this batch (the oracle restatement in float32 on the golden's variables):
    # assert_close(ref32=) holds the kernels to 1.5 x its count of elements outside the strict §8c bound
    from oracle import ref_models as M
    from tests.test_oracle_golden import _encode
    gv0 = GU.golden_to_oracle_vars(name, GU.section(d, "var/"), params)
    P32 = {k: torch.from_numpy(v.copy()).float().requires_grad_(True) for k, v in gv0.items()}
    f32 = {k: (v.float() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in _encode(params, sfeats).items()}
    okw = {"dropout_masks": [m.float() for m in GU.dropout_masks(d)]} if "aux/dropout_mask_0" in d else {}
    o32p = getattr(M, oracle_name)(P32, f32, None, params, training=False)
    o32 = getattr(M, oracle_name)(P32, f32, {"read_comment": labels.float()}, params, training=True, **okw)
    o32["loss"].backward()
    g32 = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in P32.items()}
    feats = {k: (v.float() if isinstance(v, torch.Tensor) else v) for k, v in sfeats.items()}
    labels = {"read_comment": labels.float()}
    est = Estimator(model_fn, params, RunConfig(device=dev, seed=3, use_hip_graph=False))
    est.build(feats, labels)
    feats, labels = est._to_device(feats, labels)
    arrays = est.store.named_arrays()
    gv = GU.golden_to_oracle_vars(name, GU.section(d, "var/"), params)
    missing = [k for k in gv if k not in arrays and "dice_bn" not in k]
    assert not missing, f"golden (reference) variables absent from the mirror: {missing}"
    extra = [k for k in arrays if k not in gv and not re.search(r"/(wl|bl)$", k)]
    assert not extra, f"mirror variables the reference does not have: {extra}"
    for k, v in gv.items():
        if k in arrays:
            arrays[k].copy_(torch.from_numpy(v).float().reshape(arrays[k].shape))
    before = {k: v.detach().cpu().double().clone() for k, v in est.store.named_arrays().items()}
    # PREDICT
    pr = est._call_model_fn(feats, None, ModeKeys.PREDICT)
    for k, v in GU.section(d, "predict/").items():
        ok = {"probabilities": "prob"}.get(k, k)
        assert_close(pr.predictions[k], torch.from_numpy(v), what=f"{name} predict/{k}", ref32=o32p.get(ok))
    # TRAIN: loss, gradients, one TF1-Adam step
    # training-mode dropout (NFM's hard-coded one; the reference's default rate 0.1 in the *_dropout goldens): the keep masks
    # the reference run drew are part of the golden, consumed in call order
    from recalgorithm_amd import nn
    nn.DROPOUT_KEEP_MASKS[:] = GU.dropout_masks(d)
    spec = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    assert not nn.DROPOUT_KEEP_MASKS, "the mirror made fewer dropout calls than the reference"
    assert_close(spec.loss, torch.from_numpy(d["train/loss"]), what=f"{name} loss", ref32=o32["loss"])
    spec.loss.backward()
    grads = named_grads(est.store)
    gg = GU.golden_to_oracle_vars(name, GU.section(d, "grad/"), params)
    gmax = {k: float(np.abs(v).max()) for k, v in gg.items()}
    # batch-summed gradients downstream of a BatchNorm cancel (sum_b g_b = 0): their fp32 error is
    # set by the size of the terms, i.e. by the largest gradients of the dense stack
    dense_floor = 1e-6 * max(v for k, v in gmax.items() if "embedding_weights" not in k)
    for k, g in gg.items():
        if k not in grads:
            continue
        sib = k.replace("/bias", "/kernel")
        floor = dense_floor + (1e-5 * gmax[sib] if k.endswith("/bias") and sib in gmax else 0.0)
        assert_close(grads[k], torch.from_numpy(g), what=f"{name} d({k})", reduced=True, floor=floor, ref32=g32.get(k))
    spec.train_op.optimizer.apply_gradients(est.store)
    after = est.store.named_arrays()
    ga = GU.golden_to_oracle_vars(name, GU.section(d, "var_after/"), params)
    lr = float(d["meta/learning_rate"])
    for k, va in ga.items():
        if k not in after:
            continue
        ref_upd = torch.from_numpy(va).reshape(before[k].shape) - torch.from_numpy(gv[k]).reshape(before[k].shape)
        upd = after[k].detach().cpu().double() - before[k]
        if "moving_" in k:        # BatchNorm moving statistics (momentum 0.99), updated by the forward
            # 1e-7 absolute: (1 - 0.99) * batch mean, where the mean of a centred activation is an
            # analytic zero (DIN with alpha = 1)
            assert_close(upd, ref_upd, what=f"{name} {k} update", reduced=True, floor=1e-7)
            continue
        # step 1 moves by lr*g/(|g|+eps'): ill-conditioned where |g| ~ eps' — bound as in test_gpu_models
        gref = torch.from_numpy(gg[k]).reshape(before[k].shape).abs()
        tol_g = 1e-5 * (gref + gref.pow(2).mean().sqrt()) + 1e-6 * gref.max() + dense_floor + \
            (1e-5 * gmax.get(k.replace("/bias", "/kernel"), 0.0) if k.endswith("/bias") else 0.0)
        assert_adam_update(upd, ref_upd, before[k], gref, tol_g, lr, what=f"{name} adam update {k}")
