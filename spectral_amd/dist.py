"""Multi-GPU sharding of a candidate batch and the global arg-min.

Candidates are independent: rank r of G owns the contiguous shard
[r*ceil(B/G), ...) and never exchanges corridor data.  The only collective is the reduction
of one (cost, global index) pair per rank and arg-min group -- 16 bytes -- done as an
all_gather (RCCL has no MINLOC) followed by a local min; ties go to the lowest global index,
so every rank picks the same winner.  Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world, rank):
    """Contiguous shard [lo, hi) of B candidates for `rank` of `world`."""
    per = (B + world - 1) // world
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def global_argmin(best_cost, best_idx, group=None):
    """best_cost [n] float64, best_idx [n] int64 (global indices, -1 = none) of this rank.
    Returns (cost [n], idx [n]) of the winners over all ranks; identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return best_cost, best_idx
    pair = torch.stack([best_cost.to(torch.float64), best_idx.to(torch.float64)], dim=-1).contiguous()
    dev = pair.device
    if dist.get_backend(group) == "gloo" and pair.is_cuda:   # CPU dry runs of the multi-rank flow
        pair = pair.cpu()
    gathered = [torch.empty_like(pair) for _ in range(world)]
    dist.all_gather(gathered, pair, group=group)
    allp = torch.stack(gathered).to(dev)             # [world, n, 2]
    cost, idx = allp[..., 0], allp[..., 1]
    idx_key = torch.where(idx < 0, torch.full_like(idx, float("inf")), idx)
    # lexicographic min over ranks: cost first, then global index (a handful of small launches whatever the
    # world size: this runs once per step, next to a ~7 ms solve)
    out_c = cost.min(dim=0).values
    tied = cost == out_c[None]
    out_k = torch.where(tied, idx_key, torch.full_like(idx_key, float("inf"))).min(dim=0).values
    out_i = torch.where(torch.isinf(out_k), torch.full_like(out_k, -1.0), out_k).to(torch.int64)
    return out_c, out_i
