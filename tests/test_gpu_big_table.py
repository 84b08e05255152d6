"""BASELINE.json configs[4] territory on ONE GPU: an embedding arena whose byte size exceeds 2^32 (70 M rows x 16 floats =
4.48 GB; with the gradient arena, both Adam moments and `last_step` ~ 18 GB resident).  configs[4] itself is a 100 M x 16
table row-sharded over 8 GPUs (12.5 M rows per shard): this is 5.6 x that shard, and every row-address computation of
the lookup (csrc/embed.hip), the owner-computes scatter and the deferred-exact TF1 Adam (csrc/sparse.hip: catch-up,
sweep, apply, sync) is exercised on rows on BOTH sides of the 2^31- and 2^32-byte boundaries.

  * gather: bit-exact against torch's own indexing of the arena (no kernel of this repo on the reference side);
  * three optimizer steps (sweep on, default period) on the big arena == the same three steps on a SMALL arena that holds
    just the touched rows (GRAD-mode sums + the dense TF1 Adam pass over every row of the small arena,
    /root/reference algorithm/DeepFM/deepfm.py:246-250 semantics): w, m, v of every touched row bit for bit after the
    flush; what each step's forward lookup reads, bit for bit; and within fp32 rounding of an fp64 torch Adam;
  * rows no batch touched keep their initial weights and zero moments, bit for bit, on both sides of the boundaries.
"""
import ctypes

import pytest
import torch

from tests.util import assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu

ROWS, K, F, N_EX, STEPS, LR = 70_000_000, 16, 3, 2048, 3, 0.01


class _Store:
    def __init__(self, dev):
        self.opt_state = {"step": torch.zeros(1, dtype=torch.int64, device=dev), "lr_t": torch.zeros(1, device=dev)}
        self.arenas = {}


def _arena(dev, rows, name, seed=3):
    from recalgorithm_amd.variables import EmbeddingArena
    ar = EmbeddingArena(name, K, dev, seed=seed)
    ar.add_table("t0", rows)
    ar.materialize()
    return ar


def _batches(gen):
    b32 = (1 << 32) // (K * 4)            # first row whose byte offset is >= 2^32
    b31 = (1 << 31) // (K * 4)
    special = torch.tensor([0, 1, b31 - 1, b31, b31 + 1, b32 - 2, b32 - 1, b32, b32 + 1, b32 + 257, ROWS - 2, ROWS - 1])
    out = []
    for step in range(STEPS):
        ids = torch.randint(0, ROWS, (N_EX, F), generator=gen)
        window = b32 - 150 + torch.randint(0, 300, (N_EX, F), generator=gen)        # a cloud straddling the 4 GiB boundary
        ids = torch.where(torch.rand(N_EX, F, generator=gen) < 0.3, window, ids)
        tail = ROWS - 1 - torch.randint(0, 100, (N_EX, F), generator=gen)           # ... and the last rows of the arena
        ids = torch.where(torch.rand(N_EX, F, generator=gen) < 0.05, tail, ids)
        pick = special[torch.randint(0, special.numel(), (N_EX, F), generator=gen)]
        ids = torch.where(torch.rand(N_EX, F, generator=gen) < 0.1, pick, ids)       # duplicates of the boundary rows
        ids[torch.rand(N_EX, F, generator=gen) < 0.02] = -1
        ids[:special.numel(), 0] = special                                            # every special row in every step
        if step == 1:                      # rows that return after a gap (their catch-up replays a missed step)
            ids[:, 2] = torch.where(torch.rand(N_EX, generator=gen) < 0.5, ids[:, 2], torch.randint(0, ROWS, (N_EX,), generator=gen))
        out.append((ids.contiguous(), torch.randn(N_EX, F * K, generator=gen)))
    return out


def test_lookup_scatter_and_deferred_adam_beyond_4gib(dev):
    from recalgorithm_amd import _lib, ops, sparse
    free, _total = torch.cuda.mem_get_info(dev)
    if free < 40 * (1 << 30):
        pytest.skip(f"needs ~20 GB of HBM, {free >> 30} GB free")
    lib = _lib.load()
    gen = torch.Generator().manual_seed(70)
    batches = _batches(gen)
    big = _arena(dev, ROWS, "big")
    assert big.weight.numel() * 4 > (1 << 32)
    stB = _Store(dev)
    stB.arenas["big"] = big
    # the touched rows, and a small arena holding exactly them (local row = rank of the global row)
    all_ids = torch.cat([b[0].reshape(-1) for b in batches])
    touched = torch.unique(all_ids[all_ids >= 0])
    n_small = int(touched.numel())
    small = _arena(dev, n_small, "small")
    touched_d = touched.to(dev)
    w0 = big.weight[touched_d].clone()
    small.weight.copy_(w0)
    stA = _Store(dev)
    stA.arenas["small"] = small
    # untouched witnesses on both sides of the boundaries
    b32 = (1 << 32) // (K * 4)
    cand = torch.cat([torch.arange(b32 - 4000, b32 + 4000), torch.arange(ROWS - 4000, ROWS), torch.arange(0, 4000),
                      torch.randint(0, ROWS, (20000,), generator=gen)])
    keep = ~torch.isin(cand, touched)
    witness = cand[keep].to(dev)
    w_wit = big.weight[witness].clone()

    rb0 = torch.zeros(F, dtype=torch.int64, device=dev)
    pp = lambda t_: ctypes.c_void_p(t_.data_ptr())
    p64, m64, v64 = w0.cpu().double(), torch.zeros(n_small, K, dtype=torch.float64), torch.zeros(n_small, K, dtype=torch.float64)
    for step, (ids, g) in enumerate(batches, start=1):
        local = torch.where(ids >= 0, torch.searchsorted(touched, ids.clamp(min=0)), torch.full_like(ids, -1))
        ids_d, loc_d, g_d = ids.to(dev), local.to(dev), g.to(dev)
        with torch.enable_grad():
            sB = sparse.begin_lookup(big, stB, ids_d, None, None, 0, N_EX, F)         # (catches the lookup's rows up)
            sA = sparse.begin_lookup(small, stA, loc_d, None, None, 0, N_EX, F)
        assert sB is not None and sA is not None
        st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        outB, outA = torch.empty(N_EX, F * K, device=dev), torch.empty(N_EX, F * K, device=dev)
        dvB, stpB = sparse.view_for(sB, big, stB)
        _lib.check(lib.recalgo_embedding_gather_fwd_deferred(pp(ids_d), pp(big.weight), pp(rb0), N_EX, F, K, pp(outB), F * K, 0,
                                                             dvB, stpB, 0, st_), "gather (big)")
        _lib.check(lib.recalgo_embedding_gather_fwd(pp(loc_d), pp(small.weight), pp(rb0), N_EX, F, K, pp(outA), F * K, 0, st_), "gather (small)")
        assert_bit_exact(outB, outA, f"step {step}: the lookup on the 4.5 GB arena reads what the dense pass produced")
        if step == 1:                                                                   # nothing of this repo on the reference side
            want = torch.where((ids_d >= 0).unsqueeze(-1), big.weight[ids_d.clamp(min=0).reshape(-1)].reshape(N_EX, F, K),
                               torch.zeros((), device=dev))
            assert_bit_exact(outB, want.reshape(N_EX, F * K).contiguous(), "gather vs torch indexing across the 2^32-byte boundary")
        sA.set_grad(g_d)
        sB.set_grad(g_d)
        sparse.materialize_grads(stA)
        gsum = small.grad.clone()
        sparse.new_forward(stA)
        ops.adam_tf1_advance_(stA.opt_state["step"], stA.opt_state["lr_t"], LR)
        ops.adam_tf1_(small.weight.view(-1), small.grad.view(-1), small.m.view(-1), small.v.view(-1), step=-1, lr=LR,
                      lr_t_dev=stA.opt_state["lr_t"])
        stB.opt_state["step"] += 1
        sparse.apply(big, False, stB.opt_state["step"], LR, 0.9, 0.999, 1e-8)
        # the same step in fp64 torch arithmetic on the summed gradient rows
        from oracle import ref_ops as R
        R.adam_tf1_step(p64, gsum.cpu().double(), m64, v64, step, LR)
    sparse.sync_store(stB)
    for a, b, nm in ((small.weight, big.weight, "w"), (small.m, big.m, "m"), (small.v, big.v, "v")):
        assert_bit_exact(b[touched_d], a, f"70 M-row arena vs dense TF1 Adam on the touched rows: {nm}")
    assert_close(big.weight[touched_d], p64, what="touched rows vs fp64 Adam", rtol=1e-5)
    assert_close(big.m[touched_d], m64, what="touched rows' m vs fp64 Adam", rtol=1e-5)
    assert_bit_exact(big.weight[witness], w_wit, "untouched rows keep their weights")
    assert float(big.m[witness].abs().sum()) == 0.0 and float(big.v[witness].abs().sum()) == 0.0
    # every row with state is one of the touched rows (nothing was written through a wrapped address)
    n_state = int((sparse.plan_of(big).last_step > 0).sum())
    assert n_state == n_small, f"{n_state} rows carry optimizer state, {n_small} were touched"


@pytest.mark.gpu
@pytest.mark.parametrize("rows,period", [(5_000_003, 32), (5_000_003, 7), (300_001, 32), (40_000_000, 32)])
def test_one_sweep_period_reaches_every_row_of_a_large_arena_exactly(dev, rows, period):
    """The deferred-Adam sweep share of `prepare` (RECALGO_PREPARE_SWEEP) over ONE period brings EVERY row of the arena up to
    date — large arenas use coarser sweep blocks (256 << g rows), several 64-row units per workgroup and units strided over the
    launch — and brings it there exactly: (w, m, v) equal the row's g = 0 updates replayed one step at a time by the standalone
    sweep (recalgo_adam_deferred_sweep) on a copy."""
    from recalgorithm_amd import _lib, sparse as sp
    lib = _lib.load()
    K = 4
    gen = torch.Generator(device=dev).manual_seed(rows % 1000 + period)
    w = torch.randn(rows, K, device=dev, generator=gen)
    m = torch.randn(rows, K, device=dev, generator=gen) * 0.01
    v = torch.rand(rows, K, device=dev, generator=gen) * 1e-3 + 1e-6
    last = torch.ones(rows, dtype=torch.int32, device=dev)                    # every row carries state valid for step 1
    ring = torch.zeros(sp.LR_RING, device=dev)
    T = period + 1                                                            # steps 2 .. T: one whole period
    for t in range(1, T + 1):
        ring[t % sp.LR_RING] = 0.001 * (1.0 + 0.01 * t)                       # (lr_t of the steps the replay reads)
    w2, m2, v2, last2 = w.clone(), m.clone(), v.clone(), last.clone()
    step = torch.full((1,), 1, dtype=torch.int64, device=dev)
    d = sp._CDeferred(w.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), ring.data_ptr(), 0.9, 0.999, 1e-8)
    d2 = sp._CDeferred(w2.data_ptr(), m2.data_ptr(), v2.data_ptr(), last2.data_ptr(), ring.data_ptr(), 0.9, 0.999, 1e-8)
    nb = 10
    ws = torch.zeros(int(lib.recalgo_scatter_plan_workspace_bytes(256, nb, K)), dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for t in range(2, T + 1):
        step.fill_(t)
        _lib.check(lib.recalgo_scatter_prepare(None, K, ctypes.c_void_p(ws.data_ptr()), 256, nb, 0, sp.PREPARE_SWEEP, ctypes.byref(d), None,
                                               rows, 0, period, ctypes.c_void_p(step.data_ptr()), 0, st), "prepare (sweep)")
    # every row was visited exactly in the step whose share it belongs to: its state is valid for a step in 2 .. T
    assert int(last.min()) >= 2 and int(last.max()) <= T
    counts = torch.bincount(last.long(), minlength=T + 1)[2:]
    assert int(counts.sum()) == rows and int(counts.min()) > 0               # and every step of the period swept its share
    # bring both copies to step T with the standalone sweep and compare bit for bit (the replay is exact, in any grouping)
    step.fill_(T)
    for dd in (d, d2):
        _lib.check(lib.recalgo_adam_deferred_sweep(ctypes.byref(dd), K, 0, rows, ctypes.c_void_p(step.data_ptr()), 0, st), "sweep")
    assert int(last.min()) == T == int(last2.min())
    assert_bit_exact(w, w2, "weights after a period of sweep shares vs one full sweep")
    assert_bit_exact(m, m2, "first moments")
    assert_bit_exact(v, v2, "second moments")
