"""Multi-GPU host logic: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The detect -> align -> embed path has no cross-frame state (reference src/face_detector.cpp:139-222,
src/face_recognizer.cpp:236-304 process one image at a time), so frames are sharded across ranks
with NO data-path collective.  The only exchange step of the whole pipeline is the 1:N match
against a row-sharded gallery: every rank scores the gathered queries against its
own gallery shard and the per-rank top-k lists (KB-scale) are all-gathered and merged.
"""
from __future__ import annotations

from typing import Tuple

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


def merge_topk(scores: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge candidate lists [Q, n] -> top-k by (score desc, global index asc); idx < 0 = empty slot.

    Same total order as the single-GPU kernel (csrc/face_kernels.hip topk_merge_kernel) and the
    oracle (orc_gallery_topk), so a sharded gallery returns exactly the single-gallery answer.
    """
    s = scores.to(torch.float64).clone()
    s[idx < 0] = -float("inf")
    # lexicographic (score desc, idx asc): stable sort by idx asc first, then by score desc
    order = torch.argsort(idx.to(torch.int64), dim=1, stable=True)
    s1, i1, raw1 = torch.gather(s, 1, order), torch.gather(idx, 1, order), torch.gather(scores, 1, order)
    order2 = torch.argsort(-s1, dim=1, stable=True)[:, :k]
    return torch.gather(raw1, 1, order2), torch.gather(i1, 1, order2)


def allgather_topk(local_scores: torch.Tensor, local_idx: torch.Tensor, k: int, group=None):
    """One all-gather of the per-rank (score, global index) lists [Q, k] and a local merge.

    Payload per rank: Q*k*(4+4) bytes (64 queries x 16 -> 8 KB): latency-bound, so scores and
    indices travel in ONE collective (indices bit-cast into the float buffer).
    """
    world = dist.get_world_size(group)
    packed = torch.cat([local_scores.to(torch.float32), local_idx.to(torch.int32).view(torch.float32)], 1).contiguous()
    out = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(out, packed, group=group)
    allp = torch.cat(out, 1)                                   # [Q, world*2k] blocks of (k scores, k idx)
    q = local_scores.shape[0]
    allp = allp.view(q, world, 2, local_scores.shape[1])
    sc = allp[:, :, 0, :].reshape(q, -1)
    ix = allp[:, :, 1, :].contiguous().view(torch.int32).reshape(q, -1)
    return merge_topk(sc, ix, k)
