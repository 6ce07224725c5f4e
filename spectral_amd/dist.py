"""Multi-GPU sharding of a candidate batch and the global arg-min.

Candidates are independent: rank r of G owns the contiguous shard
[r*ceil(B/G), ...) and never exchanges corridor data.  The only collective is the reduction
of one (cost, global index) pair per rank and arg-min group -- 16 bytes -- done as an
all_gather (RCCL has no MINLOC) followed by a local min; ties go to the lowest global index,
so every rank picks the same winner.  Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests.
"""
import datetime
import os

import torch
import torch.distributed as dist

# Read once by ROCr when the HIP runtime comes up: set it here, at import, if the launcher did not (effective only when
# nothing has initialised the GPU yet -- init_process_group warns otherwise; never re-exec a process to "fix" this).
_IPC_MODE_WAS_SET = "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
_cuda_was_initialized_at_import = torch.cuda.is_initialized()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# A collective that does not complete within this time fails instead of hanging (a rank that died, a link that does not
# come up): the failing rank exits non-zero and the launcher (torch.distributed.run) ends the others.
COLLECTIVE_TIMEOUT_S = int(os.environ.get("BTRAPZ_COLLECTIVE_TIMEOUT_S", "60"))   # (a node whose RCCL bootstrap is slower than that can raise it)


def init_process_group(backend, local_rank=0, timeout_s=COLLECTIVE_TIMEOUT_S):
    """torch.distributed.init_process_group for one process per GPU (rendezvous from the launcher's environment:
    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) with a finite timeout on every collective.  backend "nccl" is RCCL on
    ROCm: the communicator is bound to this rank's device at once (device_id), so a first-contact problem -- no xGMI
    peer access, IPC handles refused -- shows here, before the first timed step, not inside it.
    HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the rank's environment BEFORE the HIP runtime initialises (ROCr reads it
    once): this pool's host driver supports dmabuf IPC only, and RCCL's peer buffers fail with `hipIpcGetMemHandle:
    invalid argument` without it.  The image exports it; this module sets it at import when it is absent (below), which
    helps only if nothing has touched the GPU yet -- so a process that arrives here with the runtime up and the
    variable missing is told, instead of being left to fail inside its first collective."""
    if backend == "nccl" and not _IPC_MODE_WAS_SET and torch.cuda.is_initialized() and _cuda_was_initialized_at_import:
        import warnings
        warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY was not in the environment when the HIP runtime initialised: on hosts whose "
                      "driver supports dmabuf IPC only, RCCL's peer buffers will fail (hipIpcGetMemHandle: invalid argument). "
                      "Export HSA_ENABLE_IPC_MODE_LEGACY=0 in the launcher.")
    timeout = datetime.timedelta(seconds=timeout_s)
    if backend == "nccl":
        dist.init_process_group("nccl", timeout=timeout, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, timeout=timeout)


def _all_gather_records(rec, group=None):
    """rec [n][w] int64 of every rank -> [world][n][w], by ONE all_gather_into_tensor on a preallocated buffer on rec's
    device (no Python list of tensors, no torch.stack; under nccl no host hop).  gloo dry runs of device tensors go
    through the host, as gloo has no device transport here."""
    world = dist.get_world_size(group)
    dev = rec.device
    src = rec.cpu() if (dist.get_backend(group) == "gloo" and rec.is_cuda) else rec
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src.contiguous(), group=group)     # raises on a size mismatch: same n on every rank
    return out.view((world,) + tuple(src.shape)).to(dev)


def shard_bounds(B, world, rank):
    """Contiguous shard [lo, hi) of B candidates for `rank` of `world`."""
    per = (B + world - 1) // world
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def global_argmin(best_cost, best_idx, group=None, ctx=None, force_collective=False):
    """best_cost [n] float64, best_idx [n] int64 (global candidate indices, -1 = none) of this rank.
    Returns (cost [n], idx [n]) of the winners over all ranks; identical on every rank.

    Layout contract: entry g of EVERY rank must belong to the same arg-min group, i.e. each of the n groups is spread
    over the ranks and every rank reports its local winner of it (BASELINE configs 3/4: n = 1, the whole batch sharded
    contiguously).  Groups that live entirely on one rank -- config 5 sharded by agent, 16 agents per GPU -- need no
    collective at all: their local winner IS the winner; do not call this for them (merging rank r's group g with rank
    0's group g would merge unrelated agents).  The entry count must be the same on every rank (checked).

    One all_gather of n x 16 bytes per rank: the cost travels as its float64 bit pattern next to the int64 index in one
    int64 tensor, so indices are exact (no float round trip) and there is a single collective per step.  With ctx (a
    spectral_amd.native.Context) and device tensors the reduction over the ranks is one launch of the library
    (btrapz_argmin_pairs_device) on torch's current stream; otherwise a handful of torch ops with the same result."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):   # (force_collective: the first-contact test runs the collective at one rank)
        return best_cost, best_idx
    assert best_cost.shape == best_idx.shape and best_cost.dim() == 1
    pair = torch.stack([best_cost.to(torch.float64).view(torch.int64), best_idx.to(torch.int64)], dim=-1).contiguous()
    dev = pair.device
    allp = _all_gather_records(pair, group)                  # [world, n, 2] int64
    if ctx is not None and allp.is_cuda:
        n = allp.shape[1]
        out_c = torch.empty(n, dtype=torch.float64, device=dev); out_i = torch.empty(n, dtype=torch.int64, device=dev)
        ctx.argmin_pairs_device(world, n, allp.contiguous(), out_c, out_i, stream=torch.cuda.current_stream(dev).cuda_stream)
        return out_c, out_i
    cost = allp[..., 0].contiguous().view(torch.float64)
    idx = allp[..., 1]
    big = torch.iinfo(torch.int64).max
    idx_key = torch.where(idx < 0, torch.full_like(idx, big), idx)
    # lexicographic min over ranks: cost first (NaN never wins), then the lowest global index
    cost = torch.where(torch.isnan(cost), torch.full_like(cost, float("inf")), cost)
    out_c = cost.min(dim=0).values
    tied = cost == out_c[None]
    out_k = torch.where(tied, idx_key, torch.full_like(idx_key, big)).min(dim=0).values
    out_i = torch.where(out_k == big, torch.full_like(out_k, -1), out_k)
    return out_c, out_i


def global_argmin_with_winner(best_cost, best_idx, local_ctrl, group=None, ctx=None, force_collective=False):
    """global_argmin that also brings the winners' control points to every rank ("only the winner's 1.9 KB
    is fetched from its owner"), in the SAME collective: every rank contributes, per arg-min group, its local winner's
    (cost, global index) pair followed by that candidate's control points -- 16 B + 12 S x 8 B = 1 936 B at 20 segments
    -- and after the one all_gather every rank holds G such records, reduces the pairs as global_argmin does and keeps
    the record of the rank that owns the winner.  No second collective, and no host round trip to learn the owner (a
    broadcast from the owner needs its rank on the host, i.e. a device-to-host copy and a synchronisation per step;
    `fetch_winner` below is that form, for callers that have the index on the host anyway).

    best_cost [n] float64, best_idx [n] int64 (global indices, -1 = none), local_ctrl [n][P] float64: control points of
    this rank's winner of every group (any finite filler where best_idx is -1).  Returns (cost [n], idx [n],
    ctrl [n][P]); ctrl rows of groups nobody solved are NaN.  Identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    assert best_cost.shape == best_idx.shape and best_cost.dim() == 1 and local_ctrl.dim() == 2 and local_ctrl.shape[0] == best_cost.shape[0]
    if world == 1 and not (force_collective and dist.is_initialized()):
        out = local_ctrl.clone()
        out[best_idx < 0] = float("nan")
        return best_cost, best_idx, out
    n, P = local_ctrl.shape
    rec = torch.cat([best_cost.to(torch.float64).view(torch.int64)[:, None], best_idx.to(torch.int64)[:, None],
                     local_ctrl.to(torch.float64).contiguous().view(torch.int64)], dim=1).contiguous()   # [n][2 + P] int64
    dev = rec.device
    allr = _all_gather_records(rec, group)                                # [world][n][2 + P]
    pairs = allr[..., :2].contiguous()
    if ctx is not None and pairs.is_cuda:
        out_c = torch.empty(n, dtype=torch.float64, device=dev); out_i = torch.empty(n, dtype=torch.int64, device=dev)
        ctx.argmin_pairs_device(world, n, pairs, out_c, out_i, stream=torch.cuda.current_stream(dev).cuda_stream)
    else:
        cost = pairs[..., 0].contiguous().view(torch.float64)
        idx = pairs[..., 1]
        big = torch.iinfo(torch.int64).max
        idx_key = torch.where(idx < 0, torch.full_like(idx, big), idx)
        cost = torch.where(torch.isnan(cost), torch.full_like(cost, float("inf")), cost)
        out_c = cost.min(dim=0).values
        out_k = torch.where(cost == out_c[None], idx_key, torch.full_like(idx_key, big)).min(dim=0).values
        out_i = torch.where(out_k == big, torch.full_like(out_k, -1), out_k)
    # the owner of group g's winner: the (one) rank whose record carries the winning index
    owner = ((pairs[..., 1] == out_i[None]) & (out_i[None] >= 0)).to(torch.int64).argmax(dim=0)          # [n]
    ctrl = allr[owner, torch.arange(n, device=dev), 2:].contiguous().view(torch.float64)
    ctrl = torch.where((out_i >= 0)[:, None], ctrl, torch.full_like(ctrl, float("nan")))
    return out_c, out_i, ctrl


def fetch_winner(ctrl, win_idx, per, index_base, group=None):
    """The winner's control points from its owner by ONE broadcast.  ctrl [B_local][P]: this rank's solved
    control points; win_idx: the winner's GLOBAL index as a python int (the caller has it on the host), -1 = none;
    per: candidates per rank (shard_bounds' stride); index_base: global index of this rank's candidate 0.
    Returns a [P] tensor, identical on every rank, or None when nobody solved a candidate."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    win_idx = int(win_idx)
    if win_idx < 0:
        return None
    owner = min(world - 1, win_idx // per) if world > 1 else 0
    rank = dist.get_rank(group) if world > 1 else 0
    P = ctrl.shape[1]
    buf = ctrl[win_idx - index_base].clone() if rank == owner else torch.empty(P, dtype=ctrl.dtype, device=ctrl.device)
    if world > 1:
        if dist.get_backend(group) == "gloo" and buf.is_cuda:
            host = buf.cpu(); dist.broadcast(host, src=owner, group=group); buf = host.to(ctrl.device)
        else:
            dist.broadcast(buf, src=owner, group=group)
    return buf
