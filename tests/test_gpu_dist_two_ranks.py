"""-m gpu: TWO ranks of the row-sharded data-parallel step on the real HIP kernels, on the one GPU a
test box has.  RCCL refuses two ranks on one device, so the collectives go through
parallel.HostStagedCollectives (gloo via host memory); everything else is the production N > 1
path: arenas sharded r % 2, fixed-capacity id/row exchange with -1 padding, owner-side HIP gather, the
received gradient rows as a lookup of the shard's owner-computes plan (deferred exact Adam on the shard), dense all-reduce,
loss / N; optionally Sync-BatchNorm.
Oracle: the single-process step on the concatenated global batch (same seeds) — N ranks == 1 rank.
Also runs bench.py's N = 2 code path in the same mode (launch agreement, eager fallback, overflow
retry, JSON line)."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice(x, lo, hi):
    return x[lo:hi].contiguous() if isinstance(x, torch.Tensor) else x


def _worker(rank, port, model, capacity_factor, errq, sync_bn=False):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=WORLD)
        from recalgorithm_amd import parallel as P
        from tests.test_gpu_dist import _make
        from tests.util import assert_close
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        d = P.HostStagedCollectives(dist)

        # BatchNorm: per-rank batch statistics by default (as in any data-parallel job) — compared without it; with
        # sync_batch_norm the statistics are those of the global batch and the two ranks must equal the oracle WITH it
        ref, feats, labels = _make(model, dev, batch_norm=sync_bn)   # the 1-rank oracle, global batch
        # production order: sharded at construction (attach before the build), same initial values as the oracle
        shd, _, _ = _make(model, dev, batch_norm=sync_bn,
                          before_build=lambda e: P.attach_data_parallel(e, d, capacity_factor=capacity_factor,
                                                                        sync_batch_norm=sync_bn))
        if sync_bn:
            assert shd.store.sync_bn is not None
        B = next(iter(labels.values())).shape[0]
        lo, hi = rank * B // WORLD, (rank + 1) * B // WORLD
        f_loc = {k: _slice(v, lo, hi) for k, v in feats.items()}
        l_loc = {k: _slice(v, lo, hi) for k, v in labels.items()}
        for step in range(3):
            l0 = ref.train_step(feats, labels)
            l1 = shd.train_step(f_loc, l_loc).detach().clone()
            d.all_reduce(l1)
            assert_close(l1 / WORLD, l0, what=f"{model} mean-of-rank losses vs global loss, step {step}", rtol=2e-5)
        assert not P.exchange_overflowed(shd)
        a0, a1 = ref.store.named_arrays(), shd.store.named_arrays(gather=True)   # (tables of a sharded arena: collective gather)
        for k in a0:
            if "embedding_weights" in k or "kernel/" in k:     # arena tables: compared un-sharded below
                continue
            if re.search(r"/dense(_\d+)?/bias$", k) and any(n.startswith(k.rsplit("/", 2)[0] + "/batch_normalization") for n in a0):
                continue    # a bias ahead of a training-mode BatchNorm: zero gradient analytically, its Adam step is rounding noise
            assert_close(a1[k], a0[k], rtol=3e-4, what=f"{model} {k} after 3 steps", reduced=True)
        for name, ar in ref.store.arenas.items():
            sar = shd.store.arenas[name]
            assert sar.weight.shape[0] == (ar.weight.shape[0] - rank + WORLD - 1) // WORLD
            for what in ("weight", "m", "v"):
                full = P.unshard_arena(sar, what)
                assert_close(full, getattr(ar, what), rtol=3e-4, what=f"{model} arena {name}.{what} after 3 steps", reduced=True)
        d.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        errq.put(f"rank {rank}:\n{traceback.format_exc()}")
        raise


@pytest.mark.timeout(600)
@pytest.mark.parametrize("model,capacity_factor,sync_bn", [("dcn", 2.0, False), ("deepfm", None, False), ("deepfm", 2.0, False),
                                                           ("deepfm", 2.0, True)])
def test_two_ranks_equal_one_rank(model, capacity_factor, sync_bn):
    """sync_bn: DeepFM WITH BatchNorm under attach_data_parallel(sync_batch_norm=True) — the statistics of the global batch,
    gathered between the two BatchNorm launches — equals the single-process step on the concatenated batch (moving
    statistics, gamma / beta and everything downstream included)."""
    ctx = mp.get_context("spawn")
    errq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, model, capacity_factor, errq, sync_bn)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
    errs = []
    while not errq.empty():
        errs.append(errq.get())
    for p in procs:
        if p.is_alive():
            p.terminate()
            errs.append("worker timed out")
    assert not errs, "\n".join(errs)
    assert all(p.exitcode == 0 for p in procs)


@pytest.mark.timeout(900)
# the buckets hold DISTINCT rows (about a third of the requests of this Zipf batch): 0.1 x requests / world must overflow
# -> bench repeats at 0.2, 0.4, ... until the run is clean
@pytest.mark.parametrize("capacity_factor", ["1.5", "0.1"])
def test_bench_two_rank_code_path(capacity_factor):
    env = dict(os.environ, RECALGO_DIST_BACKEND="gloo_staged", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "4", "--batch", "512",
           "--max-vocab", "20000", "--data-batches", "3", "--no-tunable", "--no-cpu-baseline", "--no-host-fed", "--capacity-factor", capacity_factor]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 1024 and out["value"] > 0
    assert out["config"]["launch"] == "eager"                       # host-staged collectives cannot be captured
    assert "roofline" in out
    assert out["rccl_ranks"] == 2 and out["communication"]["world_size"] == 2
    assert out["communication"]["per_rank_bytes_per_step"]["all_to_all_rows_and_grads"] > 0
    if capacity_factor == "0.1":
        assert re.search(r"overflow at capacity factor 0.1", r.stderr), r.stderr[-2000:]


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher around it (the shape of the driver's scaling command): bench.py
    re-execs itself under torch.distributed.run and the line says n_gpus = 2 because two ranks ran.
    On a 1-GPU test box the two ranks share the device through the host-staged bring-up collectives."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RECALGO_DIST_BACKEND="gloo_staged", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "512", "--max-vocab", "20000",
           "--data-batches", "3", "--no-tunable", "--no-cpu-baseline", "--no-host-fed", "--no-kernel-timing", "--capacity-factor", "1.5"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 1024 and out["value"] > 0


@pytest.mark.timeout(1500)
def test_bench_eight_ranks_config5_code_path():
    """`python bench.py --gpus 8 --config5` (BASELINE.json configs[4]: DeepFM, one big row-sharded table, the id / row / gradient
    all_to_all and the dense all-reduce) through bench.py ITSELF at world 8 — the command the driver's scaling run issues — on the
    bring-up backend (8 ranks share the test box's GPU, collectives bounce through host memory; a reduced table and batch): the
    line must come from 8 ranks that all took part in the collectives, with the static exchange plan's per-link bytes in it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RECALGO_DIST_BACKEND="gloo_staged", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--config5", "--big-table-rows", "400000", "--steps", "2", "--warmup", "2",
           "--batch", "256", "--max-vocab", "20000", "--data-batches", "2", "--no-tunable", "--no-cpu-baseline", "--no-host-fed",
           "--no-kernel-timing", "--capacity-factor", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 8 * 256 and out["value"] > 0
    assert out["rccl_ranks"] == 8 and out["communication"]["world_size"] == 8
    link, rank = out["communication"]["per_link_bytes_per_step"], out["communication"]["per_rank_bytes_per_step"]
    assert link["all_to_all_rows_and_grads"] * 8 == rank["all_to_all_rows_and_grads"] and link["all_to_all_ids"] > 0
    assert "DeepFM" in out["config"]["workload"] and "r % 8" in out["config"]["parallelism"]


def test_bench_refuses_more_gpus_than_the_node_has():
    """Without the bring-up backend, asking for more GPUs than the node exposes is an error, not a 1-GPU line."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RECALGO_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr, (r.returncode, r.stderr[-1000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
