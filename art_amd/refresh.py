"""Index refresh without the pickle round trip.

The reference refreshes its index through the filesystem: every encoder rank copies its
embeddings to the host, inserts them row by row into a dict, pickles a shard, rank 0 merges the
shards into one 32 GB pickle (indexer.py:146-165, art_index.py:56-100), and the index owner
unpickles it and re-uploads everything (art_index.py:231-265).  Here the encoder output stays on
the GPU: batches are scattered by passage id straight into the next generation of the index,
which is swapped in atomically when every row has arrived.  The index ends up in id order
(row position = id - id_base), independent of the order the batches came in.

Two sinks with the store's ``add_block_data(row_id, block_embeds)`` signature:

  DeviceIndexWriter   one process owns every shard (the reference's process model)
  SpmdIndexWriter     one process per GPU, each owning the rows [lo, hi) of the id range; batches
                      are exchanged by owner with all_to_all in rounds, while the encoder runs
"""
import numpy as np
import torch
import torch.distributed as dist

from ._lib import ArtmipsError
from .distributed import shard_bounds


def _as_cuda(row_id, block_embeds, device):
    if isinstance(row_id, np.ndarray):
        row_id = torch.from_numpy(np.ascontiguousarray(row_id).astype(np.int64))
    if isinstance(block_embeds, np.ndarray):
        block_embeds = torch.from_numpy(np.ascontiguousarray(block_embeds))
    row_id = row_id.detach().to(device=device, dtype=torch.int64).reshape(-1)
    block_embeds = block_embeds.detach().to(device=device)
    if block_embeds.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        block_embeds = block_embeds.float()
    if block_embeds.dim() != 2 or block_embeds.shape[0] != row_id.shape[0]:
        raise ValueError("block_embeds must be [len(row_id), dim]")
    return row_id, block_embeds.contiguous()


def _finalize(engine):
    try:
        engine.finalize()
    except ArtmipsError as e:
        if "overwrite block data" in str(e):
            # the reference's duplicate-id error (art_index.py:59)
            raise ValueError("Unexpectedly tried to overwrite block data") from e
        raise


class DeviceIndexWriter(object):
    """Builds the next generation of a DistributedBruteForceIndex from device batches.

    index: art_amd.art_index.DistributedBruteForceIndex (searches keep hitting the live
    generation until commit()).  n_rows: number of passages; their ids must be exactly
    id_base .. id_base + n_rows - 1 (the evidence dataset uses row_id = idx + 1,
    pretokenized_evidence.py:63-64).
    """

    def __init__(self, index, n_rows, id_base=1):
        from .engine import MipsEngine

        self.index = index
        self.n_rows = int(n_rows)
        self.id_base = int(id_base)
        self._own_engine = index.evidence_embeds is None
        self.engine = index.evidence_embeds
        if self.engine is None:
            self.engine = MipsEngine(index.embed_size, index.storage_dtype, index.devices)
        self.device = torch.device("cuda", self.engine.devices[0])
        self.engine.begin(self.n_rows)

    def add_block_data(self, row_id, block_embeds, allow_overwrite=False):
        if allow_overwrite:
            raise ValueError("allow_overwrite is not supported by the device refresh path")
        row_id, block_embeds = _as_cuda(row_id, block_embeds, self.device)
        self.engine.upload_scatter(block_embeds, row_id, self.id_base)

    def commit(self, save_flat=None, background=True):
        """Swap the new generation in.  Raises ValueError if a passage id was added twice and
        ArtmipsError if some passage never arrived (the reference: art_index.py:59, indexer.py:161).

        save_flat: path of a flat store file to write from the new generation (the on-disk artefact
        the reference's indexer leaves at --embedding-path, indexer.py:156-163); by default on a
        background thread beside the training step's searches (returned as self.saver)."""
        _finalize(self.engine)
        idx = self.index
        idx.evidence_embeds = self.engine
        idx.num_rows = self.n_rows
        idx.chunksize = -(-self.n_rows // idx.ngpu)
        idx.last_chunksize = max(0, self.n_rows - idx.chunksize * (idx.ngpu - 1))
        idx.indices_arr = np.arange(self.id_base, self.id_base + self.n_rows, dtype=np.int64)
        from .art_index import _IdMap

        idx.id_map = _IdMap(idx.indices_arr)
        self.saver = idx.save_flat(save_flat, background=background) if save_flat else None
        return self.saver


def route_by_owner(row_id, block_embeds, n_rows, id_base, world):
    """Sort a batch by owning rank (torch.chunk row ranges over positions id - id_base).
    Returns (ids sorted, rows sorted, rows per destination rank as a python list)."""
    chunk = -(-n_rows // world)
    pos = row_id - id_base
    if row_id.numel() and (int(pos.min()) < 0 or int(pos.max()) >= n_rows):
        raise ValueError("passage id outside [%d, %d)" % (id_base, id_base + n_rows))
    dest = torch.div(pos, chunk, rounding_mode="floor")
    order = torch.argsort(dest, stable=True)
    counts = torch.bincount(dest, minlength=world).tolist()
    return row_id[order], block_embeds[order], counts


_DTYPE_CODES = [torch.float16, torch.bfloat16, torch.float32]


class SpmdIndexWriter(object):
    """One process per GPU: this rank's engine (one shard) receives the rows whose position
    id - id_base falls in its torch.chunk range; everything else is sent to its owner.

    STREAMING: add_block_data() buffers on the device (in the encoder's dtype) and, whenever
    `flush_bytes` have accumulated, runs one exchange ROUND: an all_to_all of the per-destination row
    counts (one small message that also carries each rank's "I am committing" flag), then all_to_alls
    of the ids and of the rows (RCCL over xGMI with the "nccl" backend), asynchronous: the transfer of
    round f runs on RCCL's stream beside whatever the encoder enqueues for round f+1, and its rows are
    scattered into the local shard when round f+1 starts (or at commit).  A rank therefore holds at
    most the batch it is accumulating plus one round in flight, not its whole output (the reference
    side this replaces keeps everything in a dict until the end and moves it through the filesystem:
    indexer.py:146-165).  With the reference's sampler consecutive ids are encoded at the same time
    (samplers.py:142-148), so at any moment ALL ranks send to the same owner: that owner receives up to
    world x flush_bytes per round.

    Rounds are rendezvous, so all ranks run the same number of them whatever their batch counts and
    sizes: a rank triggers a round when its buffer is full; commit() keeps running rounds with the
    flag set until a round in which every rank had it set.  A rank that finished early simply serves
    the others' rounds from inside commit().  Duplicate and missing rows are detected by the engine
    over the whole generation (scatter atomics + the count at finalize), across all rounds.

    `sink(rows, ids, id_base_local)` defaults to engine.upload_scatter and exists so that the
    exchange can be tested on the CPU with gloo.
    """

    def __init__(self, engine, n_rows, id_base=1, group=None, sink=None, device=None, flush_bytes=64 << 20,
                 always_exchange=False):
        self.engine = engine
        self.n_rows = int(n_rows)
        self.id_base = int(id_base)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi = shard_bounds(self.n_rows, self.world)[self.rank]
        if device is None:
            device = torch.device("cuda", engine.devices[0])
        self.device = device
        self._sink = sink if sink is not None else engine.upload_scatter
        self.flush_bytes = int(flush_bytes)
        self._ids, self._rows = [], []
        self._acc_bytes = 0          # bytes in the batch being accumulated
        self._pending = None         # the round in flight: (works, ids_in, rows_in, keep-alive, bytes)
        self._dim = engine.dim if engine is not None else None
        self.rounds = 0              # exchange rounds run so far (identical on every rank)
        self.rows_received = 0
        self.peak_buffered_bytes = 0  # accumulating batch + round in flight, the most this rank ever held
        # run the collectives even with one rank (measurements of the exchange path on a one-GPU box)
        self._exchange = self.world > 1 or (always_exchange and dist.is_initialized())
        self._nccl = self._exchange and dist.get_backend(group) == "nccl"
        if engine is not None:
            engine.begin(self.hi - self.lo)

    def _note_peak(self):
        held = self._acc_bytes + (self._pending[4] if self._pending is not None else 0)
        if held > self.peak_buffered_bytes:
            self.peak_buffered_bytes = held

    def add_block_data(self, row_id, block_embeds, allow_overwrite=False):
        if allow_overwrite:
            raise ValueError("allow_overwrite is not supported by the device refresh path")
        if self.device.type == "cuda":
            row_id, block_embeds = _as_cuda(row_id, block_embeds, self.device)
        row_id = row_id.reshape(-1).to(torch.int64)
        self._ids.append(row_id)
        self._rows.append(block_embeds)
        if self._dim is None:
            self._dim = int(block_embeds.shape[1])
        self._acc_bytes += block_embeds.numel() * block_embeds.element_size() + row_id.numel() * 8
        self._note_peak()
        if self._acc_bytes >= self.flush_bytes:
            self._round(final=False)

    def _drain(self):
        """Finish the round in flight: wait for its transfer (the current stream waits, not the host,
        under "nccl") and scatter what arrived into the local shard."""
        if self._pending is None:
            return
        works, ids_in, rows_in, _keep, _bytes = self._pending
        for w in works:
            if w is not None:
                w.wait()
        self._pending = None
        if ids_in.numel():
            ids_in, rows_in = ids_in.to(self.device), rows_in.to(self.device)
            self._sink(rows_in, ids_in, self.id_base + self.lo)
            self.rows_received += int(ids_in.numel())

    def _round(self, final):
        """One exchange round (collective).  Returns True when every rank passed final=True."""
        dim = self._dim if self._dim is not None else 0
        if self._ids:
            ids = torch.cat(self._ids)
            rows = torch.cat(self._rows)
        else:
            ids = torch.empty(0, dtype=torch.int64, device=self.device)
            rows = torch.empty((0, dim), dtype=torch.float16, device=self.device)
        self._ids, self._rows = [], []
        ids, rows, send = route_by_owner(ids, rows, self.n_rows, self.id_base, self.world)
        all_final = bool(final)
        if self._exchange:
            # "nccl" (RCCL) exchanges device buffers directly; gloo (CPU tests, or two ranks sharing
            # one GPU in the test-suite) goes through host copies and runs synchronously
            xdev = self.device if self._nccl else torch.device("cpu")
            ids, rows = ids.to(xdev), rows.contiguous().to(xdev)
            dcode = _DTYPE_CODES.index(rows.dtype)
            meta = torch.tensor([[c, 1 if final else 0, rows.shape[1], dcode] for c in send], dtype=torch.int64,
                                device=xdev)
            meta_in = torch.empty_like(meta)
            dist.all_to_all_single(meta_in, meta, group=self.group)
            m = meta_in.tolist()
            recv = [int(x[0]) for x in m]
            all_final = all(int(x[1]) for x in m)
            # a rank with nothing to send learns the row width and dtype from the peers that send to it
            # (all ranks encode with one model: the senders agree)
            senders = [x for x in m if int(x[0]) > 0]
            if rows.shape[0] == 0 and senders:
                rows = torch.empty((0, int(senders[0][2])), dtype=_DTYPE_CODES[int(senders[0][3])], device=xdev)
            width = rows.shape[1]
            ids_in = torch.empty(sum(recv), dtype=torch.int64, device=xdev)
            rows_in = torch.empty((sum(recv), width), dtype=rows.dtype, device=xdev)
            w1 = dist.all_to_all_single(ids_in, ids, recv, send, group=self.group, async_op=self._nccl)
            w2 = dist.all_to_all_single(rows_in, rows, recv, send, group=self.group, async_op=self._nccl)
            # the previous round's transfer ran beside everything enqueued since: finish it now
            self._drain()
            nbytes = (rows.numel() + rows_in.numel()) * rows.element_size() + (ids.numel() + ids_in.numel()) * 8
            self._pending = ((w1, w2), ids_in, rows_in, (ids, rows), nbytes)
        else:
            self._drain()
            self._pending = ((), ids, rows, None, 0)
        self._acc_bytes = 0
        self._note_peak()
        self.rounds += 1
        if not self._exchange:
            self._drain()
        return all_final

    def commit(self, save_flat=None):
        """Collective.  Runs exchange rounds until every rank is committing, scatters what is still in
        flight, and swaps the new generation in.  save_flat: path of ONE flat store file that every rank
        writes its shard into (flat_store.save_engine_flat_shared) once the new generation is live: the
        on-disk artefact the reference's indexer leaves at --embedding-path (indexer.py:156-163)."""
        while not self._round(final=True):
            pass
        self._drain()
        if self.engine is not None:
            _finalize(self.engine)
            if save_flat:
                from .flat_store import save_engine_flat_shared

                barrier = (lambda: dist.barrier(group=self.group)) if self.world > 1 else (lambda: None)
                save_engine_flat_shared(self.engine, save_flat, self.lo, self.n_rows, self.rank, barrier)
        return self.rows_received


__all__ = ["DeviceIndexWriter", "SpmdIndexWriter", "route_by_owner"]
