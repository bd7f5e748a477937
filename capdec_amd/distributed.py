"""Caption-batch data parallelism (the only parallel dimension of the path, SURVEY.md section 8
row E): rank r decodes the contiguous block [r*ceil(N/R), (r+1)*ceil(N/R)) of the embedding
matrix with replicated weights; the single exchange is an all-gather of int32 token ids /
lengths (+ fp32 beam scores) over RCCL (torch.distributed backend "nccl"; "gloo" in CPU tests).
Rank order preserves caption order, so the gathered matrix equals the 1-GPU result."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def shard_size(n: int, world: int) -> int:
    return (n + world - 1) // world


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """all-gather row blocks produced under :func:`shard_bounds`: every rank pads its block to
    ceil(N/R) rows, one all_gather_into_tensor moves them, the padding is cut off again."""
    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] < n_total:
            raise RuntimeError(f"gather_rows: this process holds {local.shape[0]} of {n_total} rows but no "
                               "torch.distributed process group is initialised (world > 1 needs one)")
        return local[:n_total]
    if dist.get_world_size(group) == 1 and os.environ.get("CAPDEC_FORCE_DIST") != "1":
        return local[:n_total]
    world = dist.get_world_size(group)
    per = shard_size(n_total, world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if hasattr(dist, "all_gather_into_tensor") and local.is_cuda:
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    else:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad.contiguous(), group=group)
        out = torch.cat(parts, dim=0)
    return out[:n_total]


def capi_comm_from_torch(engine, group=None) -> None:
    """Create the engine's C-ABI RCCL communicator (capdec_comm_init) for the ranks of a torch.distributed group: rank 0
    makes the id, torch broadcasts its 128 bytes.  (A host without torch distributes the id itself -- INTEGRATION.md.)"""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = engine.device if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(buf, src=0, group=group)
    engine.comm_init(rank, world, bytes(buf.cpu().tolist()))


def gather_rows_capi(engine, local: torch.Tensor, n_total: int) -> torch.Tensor:
    """:func:`gather_rows` through the C ABI's own communicator (RCCL directly, no torch.distributed in the data path)"""
    return engine.gather_rows(local, n_total)


def check_world(rank: int, world: int, group=None) -> None:
    """rank / world handed to the sharded drivers must describe the process group this process is really in"""
    if world == 1 and rank == 0:
        return
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError(f"rank={rank} world={world} but torch.distributed is not initialised")
    if dist.get_world_size(group) != world or dist.get_rank(group) != rank:
        raise RuntimeError(f"rank={rank} world={world} do not match the process group "
                           f"(rank {dist.get_rank(group)} of {dist.get_world_size(group)})")


def gather_ids(ids: torch.Tensor, lens: torch.Tensor, n_total: int, scores: Optional[torch.Tensor] = None,
               group=None):
    """token ids [n_local, (beam,) T] int32 + lens [n_local(, beam)] (+ scores) -> global."""
    g_ids = gather_rows(ids, n_total, group)
    g_lens = gather_rows(lens, n_total, group)
    g_scores = gather_rows(scores, n_total, group) if scores is not None else None
    return g_ids, g_lens, g_scores
