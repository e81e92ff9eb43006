#!/usr/bin/env python
"""Generate tests/golden/fx_tokens/ (run in the build container, needs /root/reference).

The token datasets are written AND read back by the reference's own
megatron/data/indexed_dataset.py (MMapIndexedDatasetBuilder / MMapIndexedDataset), imported by path
with a stub `megatron` module (print_rank_0) and the `np.float` alias that numpy 2 removed (the
reference pins numpy < 1.24).  That pins the .idx/.bin byte format and the reader's results.
The expected top-k structure follows the loop of art_model.py:367-381 run over the reference's
reader objects.  Only data files and JSON are written; none of the reference's source.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "fx_tokens")
REF = "/root/reference/megatron/data/indexed_dataset.py"


def load_reference():
    stub = types.ModuleType("megatron")
    stub.print_rank_0 = lambda *a, **k: None
    saved = sys.modules.get("megatron")
    sys.modules["megatron"] = stub
    if not hasattr(np, "float"):
        np.float = float  # alias removed in numpy 1.24; the reference's dtype table names it
    try:
        spec = importlib.util.spec_from_file_location("ref_indexed_dataset", REF)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        if saved is None:
            sys.modules.pop("megatron", None)
        else:
            sys.modules["megatron"] = saved
    return ref


def main():
    if not os.path.exists(REF):
        print("reference not present, skipped")
        return
    ref = load_reference()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(77))
    n_items = 48
    specs = {  # name -> (dtype, vocab, max length)
        "passages_bert": (np.uint16, 30522, 120),
        "title_bert": (np.uint16, 30522, 12),
        "passages_t0": (np.int32, 32100, 140),
        "title_t0": (np.int32, 32100, 16),
    }
    readers = {}
    for name, (dt, vocab, max_len) in specs.items():
        prefix = os.path.join(OUT, name)
        b = ref.MMapIndexedDatasetBuilder(prefix + ".bin", dtype=dt)
        for i in range(n_items):
            ln = 0 if (i == 5 and name.startswith("title")) else int(rng.integers(1, max_len + 1))
            b.add_item(torch.from_numpy(rng.integers(0, vocab, size=ln).astype(np.int64)))
            b.end_document()
        b.finalize(prefix + ".idx")
        readers[name] = ref.MMapIndexedDataset(prefix, skip_warmup=True)
    # what the reference's reader returns for every item
    expected = {}
    for name, ds in readers.items():
        expected[name + "_sizes"] = np.asarray([len(ds[i]) for i in range(len(ds))], dtype=np.int32)
        expected[name + "_flat"] = np.concatenate([np.asarray(ds[i]).astype(np.int64) for i in range(len(ds))])
    # art_model.py:367-381 over a [Q, k] id matrix (ids are 1-based, looked up at idx - 1)
    topkindex = rng.integers(1, n_items + 1, size=(3, 5)).astype(np.int32)
    topkindex[0, 0] = 6  # the item with an empty title
    topk_data = []
    for topkarray in topkindex:
        topkarray = topkarray.tolist()
        bert, t0 = [], []
        for idx in topkarray:
            bert.append((readers["passages_bert"][idx - 1].tolist(), readers["title_bert"][idx - 1].tolist()))
            t0.append((readers["passages_t0"][idx - 1].tolist(), readers["title_t0"][idx - 1].tolist()))
        topk_data.append((topkarray, bert, t0))
    np.savez(os.path.join(OUT, "expected.npz"), topkindex=topkindex, **expected)
    with open(os.path.join(OUT, "topk_data.json"), "w") as f:
        json.dump(topk_data, f)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
