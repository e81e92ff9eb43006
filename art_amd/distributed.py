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


def all_gather_rows(out, inp, group=None, async_op=False):
    """All-gather equal-size contiguous tensors: out [world, *inp.shape] <- every rank's inp.

    "nccl" (RCCL over xGMI) gathers device buffers in place, one collective.  Any other backend is a
    test backend here (gloo: CPU tests, or several ranks sharing the one GPU of the test box) and is
    kept away from device memory: the payload is staged through host tensors explicitly, because
    gloo's own CUDA-tensor path intermittently deadlocks when two processes share a GPU (seen on the
    MI355X test box with nothing but torch.zeros + dist.all_gather in the loop)."""
    if dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(out, inp, group=group, async_op=async_op)
    world = dist.get_world_size(group)
    host = inp.detach().cpu().contiguous()
    pieces = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(pieces, host, group=group)
    out.copy_(torch.stack(pieces, dim=0).view(out.shape))
    return None


def exchange_candidates(scores, ids, group=None):
    """All-gather per-rank [Q, k] (fp32 score, int32 id) lists into [world, Q, k] pairs with a
    single collective: the two arrays travel as one int32 tensor [Q, k, 2]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return scores.unsqueeze(0), ids.unsqueeze(0)
    packed = torch.stack((scores.contiguous().view(torch.int32), ids), dim=-1).contiguous()
    allp = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
    all_gather_rows(allp, packed, group)
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


class SpmdSearcher(object):
    """The SPMD step of a one-shard MipsEngine with one collective and one host wait:

        search_local_async (no wait) -> all_gather_into_tensor of the packed list -> merge_packed_sync
        (merge + ONE stream wait + ONE int32 flag: the OR of every rank's overflow bit, identical on
        all ranks, written by the merge kernel into a pinned host word).

    Everything up to the flag read is only enqueued, so the launch latency of the collective and of
    the merge hides behind the scan.  If some rank's candidate lists overflowed (rare: the lists
    are sized 8x the expectation) every rank sees it in the flag, grows its lists and repeats the
    step, so the ranks never disagree about the number of collectives.
    Buffers are cached per (Q, k): the steady state allocates only the two result tensors.
    `engine` is a MipsEngine, or anything with its search_local_async / merge_packed_sync / grow_lists
    (the CPU tests of the collective protocol drive a stand-in that injects overflow flags)."""

    def __init__(self, engine, group=None, always_gather=False, chunk_queries=2048):
        if chunk_queries < 256 or chunk_queries % 256:
            raise ValueError("chunk_queries must be a positive multiple of 256 (queries per index sweep)")
        self.chunk_queries = int(chunk_queries)
        self.engine = engine
        self.group = group
        self.always_gather = always_gather  # run the collective even with one rank (measurements)
        self._bufs = {}
        self.last_retries = 0
        self._world = None
        self._nccl = False

    def _buffers(self, n_q, top_k, world, dev):
        key = (n_q, top_k, world)
        b = self._bufs.get(key)
        if b is None:
            from .engine import MipsEngine

            words = MipsEngine.packed_words(n_q, top_k)
            gathered = torch.empty((world, words), dtype=torch.int32, device=dev)
            alias = (world == 1 and not self.always_gather)
            local = gathered[0] if alias else torch.empty((words,), dtype=torch.int32, device=dev)
            b = (local, gathered)
            if len(self._bufs) >= 8:
                self._bufs.clear()
            self._bufs[key] = b
        return b

    def search(self, queries, top_k, score_dtype=torch.float16):
        if self._world is None:  # the group does not change under a searcher: ask once
            self._world = dist.get_world_size(self.group) if dist.is_initialized() else 1
            self._nccl = dist.is_initialized() and dist.get_backend(self.group) == "nccl"
        world = self._world
        n_q = queries.shape[0]
        if n_q > self.chunk_queries and (world > 1 or self.always_gather):
            return self._search_pipelined(queries, int(top_k), score_dtype)
        local, gathered = self._buffers(n_q, int(top_k), world, queries.device)
        self.last_retries = 0
        while True:
            self.engine.search_local_async(0, queries, top_k, local)
            if world > 1 or self.always_gather:
                self._all_gather(gathered, local)
            # the step's only host wait: the merge writes the flag into a pinned host word and the
            # stream is synchronised inside the call (no device-to-host copy behind the merge)
            s, i, status = self.engine.merge_packed_sync(gathered, n_q, top_k, score_dtype)
            if status & 2:
                raise RuntimeError("artmips: internal key scratch too small on some rank")
            if not (status & 1):
                return s, i
            self.engine.grow_lists()
            self.last_retries += 1

    def _all_gather(self, gathered, local, async_op=False):
        return all_gather_rows(gathered, local, self.group, async_op=async_op and self._nccl)

    def _search_pipelined(self, queries, top_k, score_dtype):
        """Large batches (evaluation sets, BASELINE configs 3 and 4): the queries are cut into chunks
        of `chunk_queries` (a multiple of the 256 queries one sweep of the shard scores), and the
        all-gather of chunk c runs on RCCL's stream while the scan of chunk c+1 runs on the search
        stream, so only the last chunk's exchange is exposed (at 16,384 x top-1000 every rank
        receives 131 MB from each peer: milliseconds over xGMI against ~10 ms of scan per chunk).
        The merges follow in stream order; one overflow flag per chunk, ORed."""
        from .engine import MipsEngine

        world = self._world
        n_q = queries.shape[0]
        bounds = [(q0, min(q0 + self.chunk_queries, n_q)) for q0 in range(0, n_q, self.chunk_queries)]
        dev = queries.device
        key = ("pipe", n_q, top_k, world)
        bufs = self._bufs.get(key)
        if bufs is None:
            if len(self._bufs) >= 8:
                self._bufs.clear()
            bufs = []
            for q0, q1 in bounds:
                words = MipsEngine.packed_words(q1 - q0, top_k)
                bufs.append((torch.empty((words,), dtype=torch.int32, device=dev),
                             torch.empty((world, words), dtype=torch.int32, device=dev)))
            self._bufs[key] = bufs
        out_s = torch.empty((n_q, top_k), dtype=score_dtype, device=dev)
        out_i = torch.empty((n_q, top_k), dtype=torch.int32, device=dev)
        self.last_retries = 0
        while True:
            works = []
            for (q0, q1), (local, gathered) in zip(bounds, bufs):
                self.engine.search_local_async(0, queries[q0:q1], top_k, local)
                works.append(self._all_gather(gathered, local, async_op=True))
            status = 0
            for (q0, q1), (local, gathered), work in zip(bounds, bufs, works):
                if work is not None:
                    work.wait()  # the search stream waits for the collective; the host does not
                s, i, st = self.engine.merge_packed_sync(gathered, q1 - q0, top_k, score_dtype)
                out_s[q0:q1] = s
                out_i[q0:q1] = i
                status |= st
            if status & 2:
                raise RuntimeError("artmips: internal key scratch too small on some rank")
            if not (status & 1):
                if dev.type == "cuda":
                    torch.cuda.current_stream(dev).synchronize()
                return out_s, out_i
            self.engine.grow_lists()
            self.last_retries += 1


def engine_spmd_search(engine, queries, top_k, score_dtype=torch.float16, group=None):
    """SPMD search of a one-shard MipsEngine (one SpmdSearcher per engine, cached on it)."""
    searcher = getattr(engine, "_spmd_searcher", None)
    if searcher is None or searcher.group is not group:
        searcher = SpmdSearcher(engine, group)
        engine._spmd_searcher = searcher
    return searcher.search(queries, top_k, score_dtype)
