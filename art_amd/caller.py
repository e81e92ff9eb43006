"""Caller-side protocol of the search, SPMD form.

The reference gathers every rank's queries, searches on ONE rank per node and broadcasts the
[Q, k] results twice (``megatron/model/art_model.py:345-365``,
``tasks/dense_retriever/supervised_training/evaluation/evaluate.py:140-177``).  With one row shard
per rank (``art_amd.distributed.SpmdSearcher``) every rank searches its own shard with the gathered
queries and already holds the merged result: the two broadcasts and the node-first-rank
serialisation disappear.  The functions below return exactly what the reference's code holds after
its ``torch.split(...)[local_rank]`` line: this rank's rows of ``distance`` (fp16) and ``topkindex``
(int32) on the current device.
"""
import torch
import torch.distributed as dist

from .distributed import all_gather_rows


def _world(group):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group):
    return dist.get_rank(group) if dist.is_initialized() else 0


def gather_queries(query_tensor, group=None):
    """Equal-size gather of every rank's [b, dim] queries into [world*b, dim], rank order
    (art_model.py:346-353: all_gather into a list, then torch.cat)."""
    world = _world(group)
    q = query_tensor.detach().contiguous()
    if world == 1:
        return q
    out = torch.empty((world * q.shape[0],) + tuple(q.shape[1:]), dtype=q.dtype, device=q.device)
    all_gather_rows(out.view((world,) + tuple(q.shape)), q, group)
    return out


def varsize_gather_nograd(x, group=None):
    """Gather tensors whose first dimension differs between ranks; returns (cat of the unpadded
    pieces in rank order, list of per-rank sizes).  Same result as evaluate.py:338-360 (whose
    ``allsizes`` is a list of 1-element tensors; plain ints here index and split the same way)."""
    world = _world(group)
    x = x.detach().contiguous()
    if world == 1:
        return x, [int(x.shape[0])]
    size = torch.tensor([x.shape[0]], device=x.device, dtype=torch.int64)
    sizes = torch.zeros((world, 1), device=x.device, dtype=torch.int64)
    all_gather_rows(sizes, size, group)
    allsizes = [int(v) for v in sizes.view(-1).tolist()]
    max_size = max(allsizes)
    padded = torch.zeros((max_size,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    padded[:x.shape[0]] = x
    pieces = torch.empty((world,) + tuple(padded.shape), dtype=x.dtype, device=x.device)
    all_gather_rows(pieces, padded, group)
    return torch.cat([pieces[r, :n] for r, n in enumerate(allsizes)], dim=0), allsizes


def spmd_get_topk(searcher, query_tensor, topk, group=None):
    """The search part of ``PreComputedEvidenceDocsRetriever.get_topk`` (art_model.py:345-365).

    query_tensor: this rank's [local_b, dim] query embeddings (same local_b on every rank).
    Returns (distance [local_b, topk] fp16, topkindex [local_b, topk] int32): this rank's rows,
    what the reference has after ``torch.split(distance, local_bsize)[self.local_rank]``."""
    local_b = query_tensor.shape[0]
    all_q = gather_queries(query_tensor, group)
    distance, topkindex = searcher.search(all_q, topk, score_dtype=torch.float16)
    r = _rank(group)
    return distance[r * local_b:(r + 1) * local_b], topkindex[r * local_b:(r + 1) * local_b]


def spmd_evaluate_search(searcher, query_tensor, top_k, group=None, shard_size=None):
    """The search section of ``OpenRetrievalEvaluator.evaluate`` (evaluate.py:140-177).

    Ranks may hold different numbers of queries.  ``shard_size`` None searches all gathered queries
    in one call (the index is swept ceil(Q/256) times); an int reproduces the reference's chunking
    (``--shard-size``, one call per chunk) for like-for-like timing.
    Returns (distance [n_local, top_k] fp16, topkindex [n_local, top_k] int32, allsizes)."""
    all_q, allsizes = varsize_gather_nograd(query_tensor, group)
    if all_q.shape[0] == 0:
        raise ValueError("no queries on any rank")
    if shard_size is None:
        distance, topkindex = searcher.search(all_q, top_k, score_dtype=torch.float16)
    else:
        ds, ix = [], []
        for i in range(0, all_q.shape[0], int(shard_size)):
            d, t = searcher.search(all_q[i:i + int(shard_size)], top_k, score_dtype=torch.float16)
            ds.append(d)
            ix.append(t)
        distance, topkindex = torch.cat(ds, dim=0), torch.cat(ix, dim=0)
    r = _rank(group)
    lo = sum(allsizes[:r])
    return distance[lo:lo + allsizes[r]], topkindex[lo:lo + allsizes[r]], allsizes


__all__ = ["gather_queries", "varsize_gather_nograd", "spmd_get_topk", "spmd_evaluate_search"]
