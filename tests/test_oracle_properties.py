"""Property sweeps (Hypothesis, derandomised) over the oracle's optimizer restatements — CPU only."""
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import ref_ops as R

SWEEP = settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))


@SWEEP
@given(rows=st.integers(1, 40), K=st.integers(1, 9), steps=st.integers(1, 4), seed=st.integers(0, 4))
def test_lazy_adam_on_all_rows_is_dense_adam(rows, K, steps, seed):
    """LazyAdamOptimizer whose slices name every row exactly once per step IS tf.train.AdamOptimizer (dien.py:328 vs
    deepfm.py:246-250): the two restatements must agree to rounding."""
    gen = torch.Generator().manual_seed(seed * 97 + rows * 11 + K)
    p0 = torch.randn(rows, K, generator=gen, dtype=torch.float64)
    pd, md, vd = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pl, ml, vl = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in range(1, steps + 1):
        g = torch.randn(rows, K, generator=gen, dtype=torch.float64)
        perm = torch.randperm(rows, generator=gen)
        R.adam_tf1_step(pd, g, md, vd, step, 0.01)
        R.lazy_adam_step(pl, perm, g[perm], ml, vl, step, 0.01)
        assert torch.allclose(pl, pd, rtol=1e-12, atol=1e-14)
        assert torch.allclose(ml, md, rtol=1e-12, atol=1e-14) and torch.allclose(vl, vd, rtol=1e-12, atol=1e-14)


@SWEEP
@given(rows=st.integers(2, 40), K=st.integers(1, 6), n=st.integers(1, 60), seed=st.integers(0, 4))
def test_lazy_adam_touches_exactly_the_named_rows_and_sums_duplicates(rows, K, n, seed):
    """_apply_sparse_duplicate_indices: duplicates are summed first, so the update of a row equals the dense Adam update with
    the summed gradient; rows that are not named keep w, m, v bit for bit."""
    gen = torch.Generator().manual_seed(seed * 89 + rows * 7 + n)
    p = torch.randn(rows, K, generator=gen, dtype=torch.float64)
    m, v = torch.rand(rows, K, generator=gen, dtype=torch.float64), torch.rand(rows, K, generator=gen, dtype=torch.float64)
    idx = torch.randint(0, max(rows // 2, 1), (n,), generator=gen)            # the upper half is never named
    vals = torch.randn(n, K, generator=gen, dtype=torch.float64)
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    R.lazy_adam_step(p, idx, vals, m, v, 3, 0.02)
    named = torch.zeros(rows, dtype=torch.bool)
    named[idx] = True
    assert torch.equal(p[~named], p0[~named]) and torch.equal(m[~named], m0[~named]) and torch.equal(v[~named], v0[~named])
    gsum = torch.zeros(rows, K, dtype=torch.float64).index_add_(0, idx, vals)
    pd, md, vd = p0.clone(), m0.clone(), v0.clone()
    R.adam_tf1_step(pd, gsum, md, vd, 3, 0.02)
    assert torch.allclose(p[named], pd[named], rtol=1e-12, atol=1e-14)
    assert torch.allclose(m[named], md[named], rtol=1e-12, atol=1e-14) and torch.allclose(v[named], vd[named], rtol=1e-12, atol=1e-14)
