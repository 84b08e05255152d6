"""The on-device batch generator of bench.py's optimizer-state sweep (io/synth.py device_fresh_batches) draws from the same
distribution and hands out the same layout as the host generator (device_features)."""
import numpy as np
import pytest
import torch

from recalgorithm_amd.io import synth


def test_device_fresh_batches_layout_and_distribution():
    spec = synth.SynthSpec(n_fields=8, max_vocab=5000, seed=3)
    B, n = 2048, 6
    fresh = synth.device_fresh_batches(spec, B, torch.device("cpu"), n, seed=11)
    ref_f, ref_l, _ = synth.device_features(spec, B, torch.device("cpu"))
    assert len(fresh) == n
    f, l = fresh[2]
    assert sorted(f) == sorted(ref_f) and list(l) == list(ref_l)
    for k in f:                                   # same view geometry: GraphedTrainStep.load moves a batch with ONE copy
        assert f[k].stride() == ref_f[k].stride() and f[k].storage_offset() == ref_f[k].storage_offset()
        assert f[k].untyped_storage().nbytes() == ref_f[k].untyped_storage().nbytes()
        assert f[k].untyped_storage().data_ptr() == l["read_comment"].untyped_storage().data_ptr()
    assert not torch.equal(fresh[0][0]["userid"], fresh[1][0]["userid"])          # never repeated
    again = synth.device_fresh_batches(spec, B, torch.device("cpu"), n, seed=11)
    assert torch.equal(again[2][0]["userid"], f["userid"])                        # seeded
    # marginals: OOV fraction, label rate, and the Zipf head of one field against the host generator's
    host = np.concatenate([synth.make_id_batch(spec, B, 100 + i, sorted(spec.names))[0] for i in range(n)])
    dev = torch.cat([torch.stack([b[0][k] for k in sorted(b[0])], 1) for b in fresh]).numpy()
    assert abs((dev < 0).mean() - spec.oov_frac) < 0.004
    assert abs(np.mean([float(b[1]["read_comment"].mean()) for b in fresh]) - 0.0356) < 0.01
    j = sorted(spec.names).index("userid")
    for top in (0, 1, 2):
        ph, pd = (host[:, j] == top).mean(), (dev[:, j] == top).mean()
        assert abs(ph - pd) < 0.25 * ph + 0.005, (top, ph, pd)
    assert int(dev.max()) < max(spec.vocabs)


def test_device_fresh_batches_rejects_ragged_specs():
    with pytest.raises(ValueError):
        synth.device_fresh_batches(synth.SynthSpec(n_fields=8, with_history=True), 16, torch.device("cpu"), 1, 0)
