"""The import surface the reference's three importers of the index layer need
(megatron/indexer.py:7, megatron/model/art_model.py:12,
tasks/dense_retriever/supervised_training/evaluation/evaluate.py:15), through the one-line shim of
INTEGRATION.md §1 placed at megatron/data/art_index.py of a stand-in package."""
import importlib
import inspect
import sys
import textwrap


def test_shim_module_satisfies_the_callers_import_lines(tmp_path, monkeypatch):
    pkg = tmp_path / "megatron"
    (pkg / "data").mkdir(parents=True)
    (pkg / "__init__.py").write_text(textwrap.dedent("""
        import types
        def get_args():
            return types.SimpleNamespace(embedding_path="/nonexistent/evidence.pkl", rank=0)
        mpu = types.SimpleNamespace(is_unitialized=lambda: True, get_data_parallel_rank=lambda: 0)
    """))
    (pkg / "data" / "__init__.py").write_text("")
    (pkg / "data" / "art_index.py").write_text("from art_amd.art_index import *\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    for name in [m for m in sys.modules if m == "megatron" or m.startswith("megatron.")]:
        monkeypatch.delitem(sys.modules, name)
    try:
        _check(monkeypatch)
    finally:  # the stand-in package must not outlive this test
        for name in [m for m in sys.modules if m == "megatron" or m.startswith("megatron.")]:
            del sys.modules[name]
        importlib.invalidate_caches()


def _check(monkeypatch):
    ns = {}
    exec("from megatron.data.art_index import detach, OpenRetreivalDataStore", ns)                    # indexer.py:7
    exec("from megatron.data.art_index import OpenRetreivalDataStore, DistributedBruteForceIndex", ns)  # art_model.py:12
    exec("from megatron.data.art_index import OpenRetreivalDataStore, FaissMIPSIndex, "
         "DistributedBruteForceIndex", ns)                                                             # evaluate.py:15
    # the constructor and method signatures the callers use
    sig = inspect.signature
    assert list(sig(ns["OpenRetreivalDataStore"].__init__).parameters)[1:4] == ["embedding_path", "load_from_path", "rank"]
    assert list(sig(ns["DistributedBruteForceIndex"].__init__).parameters)[1:3] == ["embed_size", "embed_data"]
    assert list(sig(ns["FaissMIPSIndex"].__init__).parameters)[1:4] == ["embed_size", "embed_data", "use_gpu"]
    for cls in (ns["DistributedBruteForceIndex"], ns["FaissMIPSIndex"]):
        assert list(sig(cls.search_mips_index).parameters)[1:] == ["query_embeds", "top_k", "reconstruct"]
        assert sig(cls.search_mips_index).parameters["reconstruct"].default is True
        for method in ("reset_index", "update_index", "add_embed_data", "_set_mips_index"):
            assert callable(getattr(cls, method))
    store_cls = ns["OpenRetreivalDataStore"]
    assert list(sig(store_cls.add_block_data).parameters)[1:] == ["row_id", "block_embeds", "allow_overwrite"]
    for method in ("state", "clear", "load_from_file", "save_shard", "merge_shards_and_save"):
        assert callable(getattr(store_cls, method))
    # inside "megatron" the store takes its path and rank from get_args() (art_index.py:20-23)
    store = store_cls(load_from_path=False)
    assert store.embedding_path == "/nonexistent/evidence.pkl" and store.rank == 0
