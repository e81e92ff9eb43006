#!/usr/bin/env python
"""Golden vectors for the SEARCH path produced by the reference's own code.

Runs in the build container only (it needs /root/reference; the GPU box never sees it).  The reference's
`DistributedBruteForceIndex` (megatron/data/art_index.py:200-304) is imported by path with the same stub
`megatron` module the store fixture uses, and its `add_embed_data` + `search_mips_index` are EXECUTED, not
restated.  The container has no GPU and the class addresses its shards as 'cuda:i', so for the duration
of the run the device placement — and nothing else — is redirected to the CPU:

    torch.cuda.device_count()            -> G (the number of "GPUs" = row shards of the case)
    Tensor.to('cuda:i' | device='cuda:i') -> the tensor itself
    torch.zeros(..., device="cuda")      -> the same call without the device
    Tensor.cuda()                        -> the tensor itself

Every arithmetic call is the reference's own and runs in torch as written there: np.array(..., float16),
torch.chunk, torch.matmul of fp16 tensors, the copy of the slabs into the fp16 matrix C, torch.topk, the
id_map loop.  What is stored: the inputs and the (distances fp16, indices int32) the reference returned.

Two families of cases:
  * integer-valued embeddings in {-1, 0, 1}: every dot product is a small integer, exact in fp16 under any
    summation order, with many ties -> the reference's distances are exact scores and its id choice is
    ambiguous only inside groups of equal score;
  * random fp16 embeddings: the reference ranks fp16-ROUNDED scores (art_index.py:283,294), so its
    distances are the fp16 roundings of the true top-k scores and ids are unambiguous where a rounded
    score occurs once in its list.
tests/test_ref_search_fixtures.py checks the oracle against these on the CPU; tests/test_search2_gpu.py
checks the HIP engine against them on the GPU."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/megatron/data/art_index.py"
D = 768


def load_reference():
    stub = types.ModuleType("megatron")
    mpu = types.ModuleType("megatron.mpu")
    mpu.is_unitialized = lambda: True
    mpu.get_data_parallel_rank = lambda: 0
    stub.mpu = mpu
    stub.get_args = lambda: None
    saved = {k: sys.modules.get(k) for k in ("megatron", "megatron.mpu")}
    sys.modules["megatron"] = stub
    sys.modules["megatron.mpu"] = mpu
    try:
        spec = importlib.util.spec_from_file_location("ref_art_index", REF_FILE)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ref


class devices_on_cpu(object):
    """Redirect device placement to the CPU while the reference code runs (see the module docstring)."""

    def __init__(self, n_gpu):
        self.n_gpu = n_gpu

    def __enter__(self):
        self.saved = (torch.cuda.device_count, torch.Tensor.to, torch.zeros, torch.Tensor.cuda)
        orig_to, orig_zeros = torch.Tensor.to, torch.zeros

        def is_cuda_dev(d):
            return isinstance(d, str) and d.startswith("cuda")

        def to(t, *args, **kw):
            if (args and is_cuda_dev(args[0])) or is_cuda_dev(kw.get("device")):
                args = tuple(a for a in args if not is_cuda_dev(a))
                kw = {k: v for k, v in kw.items() if k != "device"}
                return orig_to(t, *args, **kw) if (args or kw) else t
            return orig_to(t, *args, **kw)

        def zeros(*args, **kw):
            if is_cuda_dev(kw.get("device")):
                kw = {k: v for k, v in kw.items() if k != "device"}
            return orig_zeros(*args, **kw)

        n = self.n_gpu
        torch.cuda.device_count = lambda: n
        torch.Tensor.to = to
        torch.zeros = zeros
        torch.Tensor.cuda = lambda t, *a, **k: t
        return self

    def __exit__(self, *exc):
        torch.cuda.device_count, torch.Tensor.to, torch.zeros, torch.Tensor.cuda = self.saved


def run_reference(ref, rows_f16, ids, queries_f16, k, n_gpu):
    """The reference's own build + search on the given rows (dict insertion order = row order)."""
    store = ref.OpenRetreivalDataStore("/tmp/_unused_ref_fixture.pkl", load_from_path=False, rank=0)
    store.add_block_data(ids.astype(np.int64), rows_f16)
    with devices_on_cpu(n_gpu):
        index = ref.DistributedBruteForceIndex(embed_size=D, embed_data=store)
        assert index.ngpu == n_gpu and len(index.evidence_embeds) == n_gpu
        dist, idx = index.search_mips_index(torch.from_numpy(queries_f16), top_k=k, reconstruct=False)
    assert dist.dtype == torch.float16 and idx.dtype == torch.int32
    return dist.numpy(), idx.numpy(), (index.chunksize, index.last_chunksize)


def main():
    if not os.path.exists(REF_FILE):
        print("reference not present: nothing generated")
        return
    ref = load_reference()
    out = {}
    # ---- integer-valued, many ties, ragged shard split
    rng = np.random.Generator(np.random.PCG64(2025))
    n, q = 1003, 9
    rows = rng.choice(np.array([-1, 0, 0, 0, 1], dtype=np.float32), size=(n, D)).astype(np.float16)
    rows[500] = rows[17]            # exact duplicates: equal scores for every query
    rows[1002] = rows[17]
    queries = rng.choice(np.array([-1, 0, 1], dtype=np.float32), size=(q, D)).astype(np.float16)
    ids = (rng.permutation(n) + 1).astype(np.int32)
    out.update(int_rows=rows, int_queries=queries, int_ids=ids)
    for g in (1, 2, 3):
        for k in (1, 32, 100):
            d, i, chunks = run_reference(ref, rows, ids, queries, k, g)
            out["int_g%d_k%d_dist" % (g, k)] = d
            out["int_g%d_k%d_idx" % (g, k)] = i
        out["int_g%d_chunks" % g] = np.asarray(chunks)
    # ---- random fp16 rows: the reference's fp16-rounded ranking
    rng = np.random.Generator(np.random.PCG64(2026))
    n, q = 2051, 8
    rows = (rng.standard_normal((n, D)) * 0.05).astype(np.float16)
    queries = rng.standard_normal((q, D)).astype(np.float16)
    ids = (rng.permutation(n) + 1).astype(np.int32)
    out.update(rnd_rows=rows, rnd_queries=queries, rnd_ids=ids)
    for g in (1, 4):
        for k in (32, 100):
            d, i, _ = run_reference(ref, rows, ids, queries, k, g)
            out["rnd_g%d_k%d_dist" % (g, k)] = d
            out["rnd_g%d_k%d_idx" % (g, k)] = i
    np.savez_compressed(os.path.join(HERE, "fx_ref_search.npz"), **out)
    print("fx_ref_search.npz:", len(out), "arrays;",
          "int case distinct scores per query ~", len(np.unique(out["int_g1_k100_dist"][0])),
          "of 100; rnd case duplicates inside top-100 of query 0:",
          100 - len(np.unique(out["rnd_g1_k100_dist"][0])))


if __name__ == "__main__":
    main()
