"""CPU test doubles of the three local kernels of the row exchange (HIP in production:
recalgorithm_amd/parallel.py hip_exchange_plan / hip_local_gather / hip_local_scatter_add), plain
torch ops, device-agnostic.  They are also the oracle the HIP planner is compared with on the GPU."""
import torch


def cpu_gather(shard_weight, local_rows):                     # id < 0 -> zero row
    rows = shard_weight.index_select(0, local_rows.clamp(min=0))
    return torch.where((local_rows >= 0).unsqueeze(1), rows, torch.zeros_like(rows))


def cpu_scatter_add(arena, local_rows, g):                    # id < 0 skipped
    ok = local_rows >= 0
    arena.grad.index_add_(0, local_rows[ok], g[ok])


def torch_exchange_plan(rows, world, cap, overflow, with_send_pos=False):
    """Stable bucketing by owner = row % world into `world` buckets of `cap` slots: request i of
    owner w takes the next free slot of bucket w in request order; surplus requests are dropped
    and raise the flag."""
    M, dev = rows.numel(), rows.device
    valid = rows >= 0
    owner = torch.where(valid, rows % world, torch.full_like(rows, world))      # invalid -> sentinel bucket
    order = torch.argsort(owner, stable=True)
    so = owner[order]
    counts = torch.zeros(world + 1, dtype=torch.int64, device=dev).scatter_add_(0, owner, torch.ones_like(owner))
    start = torch.cumsum(counts, 0) - counts
    slot = torch.arange(M, device=dev, dtype=torch.int64) - start[so]
    keep = (so < world) & (slot < cap)
    overflow.logical_or_(((so < world) & (slot >= cap)).any().reshape(1))
    dummy = world * cap
    dest = torch.where(keep, so * cap + slot, torch.full_like(so, dummy))
    send_local = torch.full((world * cap + 1,), -1, dtype=torch.int64, device=dev)
    send_local.scatter_(0, dest, torch.div(rows[order], world, rounding_mode="floor"))
    req_slot = torch.full((M,), -1, dtype=torch.int64, device=dev)
    req_slot[order[keep]] = dest[keep]
    if with_send_pos:
        pos = torch.full((world * cap + 1,), -1, dtype=torch.int64, device=dev)
        pos.scatter_(0, dest, order)
        pos[dummy] = -1
        return send_local[:world * cap].contiguous(), pos[:world * cap].contiguous(), req_slot
    return send_local[:world * cap].contiguous(), req_slot


def torch_dedup_rows(rows):
    """-> (unique_rows, rep): rep[i] = the smallest request index asking for rows[i] (i itself for rows < 0);
    unique_rows[i] = rows[i] where rep[i] == i, else -1.  Static shapes (stable sort + running maximum)."""
    M, dev = rows.numel(), rows.device
    ar = torch.arange(M, dtype=torch.int64, device=dev)
    if M == 0:
        return rows.clone(), ar
    srt, perm = torch.sort(rows, stable=True)
    first = torch.ones(M, dtype=torch.bool, device=dev)
    first[1:] = srt[1:] != srt[:-1]
    start = torch.cummax(torch.where(first, ar, torch.zeros_like(ar)), 0).values
    rep = torch.empty_like(ar)
    rep[perm] = perm[start]
    rep = torch.where(rows >= 0, rep, ar)
    return torch.where(rep == ar, rows, torch.full_like(rows, -1)), rep
