"""SPMD form of the search: one process per GPU, each owning one row shard.

Replaces the single-process fan-out of art_index.py:273-294 (query copies to every
GPU, [Q, N/G] score slabs copied back to GPU-0, torch.topk over [Q, N]) and the
caller-side broadcast of art_model.py:361-362 with

    local top-k on every rank -> ONE all-gather of Q*k (score, id) pairs (RCCL over
    xGMI when the backend is "nccl") -> the same on-device merge on every rank.

The exchange moves Q*k*8 bytes per rank (16 KiB at Q=64, k=32) instead of
Q*N/G*2 bytes (336 MB per GPU at N=21M).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size):
    """Row range of every rank: contiguous ceil-div chunks, the torch.chunk rule the
    reference uses for its per-GPU shards (art_index.py:251-253)."""
    chunk = -(-n_rows // world_size)
    return [(min(chunk * r, n_rows), min(chunk * (r + 1), n_rows)) for r in range(world_size)]


def exchange_candidates(scores, ids, group=None):
    """All-gather per-rank [Q, k] (fp32 score, int32 id) lists into [world, Q, k] pairs with a
    single collective: the two arrays travel as one int32 tensor [Q, k, 2]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return scores.unsqueeze(0), ids.unsqueeze(0)
    packed = torch.stack((scores.contiguous().view(torch.int32), ids), dim=-1).contiguous()
    gathered = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(gathered, packed, group=group)
    allp = torch.stack(gathered, dim=0)
    return allp[..., 0].contiguous().view(torch.float32), allp[..., 1].contiguous()


def spmd_search(local_search, merge, queries, top_k, group=None):
    """Distributed exact top-k.

    local_search(queries, top_k) -> (scores fp32 [Q,k], ids int32 [Q,k]) over THIS rank's rows,
        padded with (-inf, INT32_MAX) if the shard has fewer than k rows
        (MipsEngine.search_local);
    merge(scores [G,Q,k], ids [G,Q,k]) -> (scores [Q,k], ids [Q,k]) (MipsEngine.merge).
    Every rank passes the same queries and gets the same result.
    """
    scores, ids = local_search(queries, top_k)
    all_scores, all_ids = exchange_candidates(scores, ids, group)
    return merge(all_scores, all_ids)


def engine_spmd_search(engine, queries, top_k, score_dtype=torch.float16, group=None):
    """spmd_search bound to a one-shard MipsEngine."""
    from .engine import MipsEngine

    return spmd_search(lambda q, k: engine.search_local(0, q, k),
                       lambda s, i: MipsEngine.merge(s, i, score_dtype), queries, top_k, group)
