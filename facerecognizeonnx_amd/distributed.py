"""Multi-GPU host logic: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The detect -> align -> embed path has no cross-frame state (reference src/face_detector.cpp:139-222,
src/face_recognizer.cpp:236-304 process one image at a time), so frames are sharded across ranks
with NO data-path collective.  The only exchange step of the whole pipeline is the 1:N match
against a row-sharded gallery: every rank scores the gathered queries against its
own gallery shard, the per-rank top-k lists (KB-scale) are all-gathered in ONE collective, and the
merge runs on the GPU in the same kernel the single-GPU gallery finishes with (`fh_topk_merge_dev`).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) slice of n items for `rank` (first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gallery_shard_base(total_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Row range of the gallery shard owned by `rank`; its begin is the global index base."""
    return shard_range(total_rows, rank, world)


def allgather_queries(local_q: torch.Tensor, group=None) -> torch.Tensor:
    """All ranks contribute the same number of query rows [q, dim]; returns [world*q, dim]."""
    world = dist.get_world_size(group)
    out = [torch.empty_like(local_q) for _ in range(world)]
    dist.all_gather(out, local_q.contiguous(), group=group)
    return torch.cat(out, 0)


def merge_topk_dev(part_scores: torch.Tensor, part_idx: torch.Tensor, k: int, stream: int = 0):
    """[parts, Q, k] per-shard lists on the GPU -> overall top-k [Q, k] by (score desc, global index asc): the
    library's topk_merge_kernel behind the C ABI (no torch sort on the product path)."""
    from . import _lib
    if not part_scores.is_cuda:
        raise RuntimeError("merge_topk_dev needs the per-shard lists in HBM: the product merge is a HIP kernel")
    parts, q, kk = part_scores.shape
    ps = part_scores.to(torch.float32).contiguous()
    pi = part_idx.to(torch.int32).contiguous()
    out_s = torch.empty((q, k), dtype=torch.float32, device=ps.device)
    out_i = torch.empty((q, k), dtype=torch.int32, device=ps.device)
    if kk != k:
        raise ValueError("per-shard lists must hold k entries each")
    stream = stream or torch.cuda.current_stream(ps.device).cuda_stream
    _lib.check(_lib.lib().fh_topk_merge_dev(ps.data_ptr(), pi.data_ptr(), parts, q, k, out_s.data_ptr(), out_i.data_ptr(), stream),
               "fh_topk_merge_dev")
    return out_s, out_i


def allgather_topk(local_scores: torch.Tensor, local_idx: torch.Tensor, k: int, group=None,
                   merge: Optional[Callable] = None, comm_device: Optional[torch.device] = None):
    """One all-gather of the per-rank (score, global index) lists [Q, k], then the merge.

    Payload per rank: Q*k*(4+4) bytes (64 queries x 16 -> 8 KB): latency-bound, so scores and indices travel
    in ONE collective (indices bit-cast into the float buffer).  `comm_device`: where the collective runs
    (the tensors' own device for nccl = RCCL; cpu for gloo).  `merge(part_scores, part_idx, k)` defaults to
    the GPU kernel (`merge_topk_dev`); the CPU tests pass their own checker, the product never does.
    """
    world = dist.get_world_size(group)
    q = local_scores.shape[0]
    src_dev = local_scores.device
    packed = torch.cat([local_scores.to(torch.float32), local_idx.to(torch.int32).view(torch.float32)], 1).contiguous()
    if comm_device is not None:
        packed = packed.to(comm_device)
    gathered = torch.empty((world * q, 2 * k), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(gathered, packed, group=group)           # rank-major concatenation
    gathered = gathered.to(src_dev).view(world, q, 2 * k)
    part_s = gathered[:, :, :k].contiguous()
    part_i = gathered[:, :, k:].contiguous().view(torch.int32)
    return (merge or merge_topk_dev)(part_s, part_i, k)
