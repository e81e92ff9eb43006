"""The allocation-free entry point (artmips_search_local_noalloc, include/artmips.h) against the entry
point it is a variant of (artmips_search_local_async), through the C ABI: bit-identical packed lists on
two calls in a row, the second with a smaller input, so that whatever the first call left in the static
scratch array must not leak into the second.  Also: a shape whose workspace exceeds the array is
refused before anything runs, and torch sees no allocation during the calls."""
import numpy as np
import pytest
import torch

from oracle import mips_oracle as orc

pytestmark = pytest.mark.gpu


def _case(n, q, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = (rng.standard_normal((n, 768)) * 0.05).astype(np.float16)
    queries = rng.standard_normal((q, 768)).astype(np.float16)
    ids = (rng.permutation(n) + 1).astype(np.int32)
    return rows, queries, ids


def test_noalloc_matches_async_twice_second_smaller():
    from art_amd.engine import MipsEngine
    from tests import gpu_util as gu

    assert torch.cuda.is_available()
    rows, queries, ids = _case(200000, 64, seed=501)
    eng = gu.build_engine(rows, ids)
    rows2, queries2, ids2 = _case(30000, 7, seed=502)
    eng2 = gu.build_engine(rows2, ids2)
    for engine, qs, k, r, i in ((eng, queries, 32, rows, ids), (eng2, queries2, 10, rows2, ids2)):
        q = torch.from_numpy(qs).cuda()
        words = MipsEngine.packed_words(q.shape[0], k)
        want = torch.empty((words,), dtype=torch.int32, device="cuda")
        engine.search_local_async(0, q, k, want)
        torch.cuda.synchronize()
        got = torch.full((words,), -7, dtype=torch.int32, device="cuda")
        before = torch.cuda.memory_stats()["allocation.all.allocated"]
        engine.search_local_noalloc(0, q, k, got)
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats()["allocation.all.allocated"] == before
        plane = q.shape[0] * k
        assert int(got[2 * plane].item()) == 0, "status word raised"
        assert torch.equal(got[:2 * plane], want[:2 * plane]), "packed (score bits | ids) differ"
        # and the list is the right answer, not merely the same one
        s = got[:plane].view(torch.float32).view(-1, k).cpu().numpy()
        ii = got[plane:2 * plane].view(-1, k).cpu().numpy()
        orc.check_topk(s, ii, qs.astype(np.float64), r.astype(np.float64), i, k, rel_tol=1e-3)
    # back to the large shape after the small one: still identical
    q = torch.from_numpy(queries).cuda()
    words = MipsEngine.packed_words(64, 32)
    a = torch.empty((words,), dtype=torch.int32, device="cuda")
    b = torch.empty((words,), dtype=torch.int32, device="cuda")
    eng.search_local_async(0, q, 32, a)
    eng.search_local_noalloc(0, q, 32, b)
    torch.cuda.synchronize()
    assert torch.equal(a[:2 * 64 * 32], b[:2 * 64 * 32])
    eng.close()
    eng2.close()


def test_noalloc_first_call_on_a_fresh_engine_allocates_nothing():
    """The allocation check proper: torch.cuda.memory_stats() only sees torch's caching
    allocator, and after search_local_async on the same engine the shard's on-demand buffers are grown
    already.  So: a FRESH engine, the allocation-free call FIRST; the library's own workspace accounting
    (artmips_info: every hipMalloc'ed scratch buffer of the handle) and the device's free memory
    (hipMemGetInfo) must not move.  Only then the allocating entry point, for the bit comparison."""
    from art_amd.engine import MipsEngine
    from tests import gpu_util as gu

    rows, queries, ids = _case(150000, 48, seed=504)
    eng = gu.build_engine(rows, ids)
    q = torch.from_numpy(queries).cuda()
    k = 24
    words = MipsEngine.packed_words(q.shape[0], k)
    got = torch.full((words,), -7, dtype=torch.int32, device="cuda")
    want = torch.empty((words,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ws_before = eng.info()["workspace_bytes"]
    assert ws_before == 0, "a fresh engine owns no search workspace"
    free_before = torch.cuda.mem_get_info()[0]
    torch_before = torch.cuda.memory_stats()["allocation.all.allocated"]
    for _ in range(2):
        eng.search_local_noalloc(0, q, k, got)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "device free memory moved during the call"
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == torch_before
    assert eng.info()["workspace_bytes"] == ws_before, "the library grew a workspace buffer"
    eng.search_local_async(0, q, k, want)
    torch.cuda.synchronize()
    assert eng.info()["workspace_bytes"] > 0, "the allocating entry point is expected to grow its buffers"
    plane = q.shape[0] * k
    assert int(got[2 * plane].item()) == 0
    assert torch.equal(got[:2 * plane], want[:2 * plane])
    eng.close()


def test_noalloc_under_graph_capture_and_replay():
    """The use INTEGRATION.md names: the call is captured into a HIP graph (nothing in it allocates,
    synchronises or changes a function attribute: those are set in artmips_create) and replayed with new
    queries in the same buffer."""
    from art_amd.engine import MipsEngine
    from tests import gpu_util as gu

    rows, queries, ids = _case(100000, 64, seed=505)
    eng = gu.build_engine(rows, ids)
    k = 32
    q_static = torch.from_numpy(queries).cuda()
    words = MipsEngine.packed_words(64, k)
    out = torch.zeros((words,), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):       # warm-up outside the capture, as torch asks for
        eng.search_local_noalloc(0, q_static, k, out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.search_local_noalloc(0, q_static, k, out)
    plane = 64 * k
    for seed in (506, 507):
        _, q2, _ = _case(8, 64, seed=seed)
        q_static.copy_(torch.from_numpy(q2).cuda())
        out.fill_(-7)
        graph.replay()
        torch.cuda.synchronize()
        assert int(out[2 * plane].item()) == 0
        s = out[:plane].view(torch.float32).view(-1, k).cpu().numpy()
        ii = out[plane:2 * plane].view(-1, k).cpu().numpy()
        orc.check_topk(s, ii, q2.astype(np.float64), rows.astype(np.float64), ids, k, rel_tol=1e-3)
    del graph
    eng.close()


def test_noalloc_refuses_what_does_not_fit():
    from art_amd._lib import ArtmipsError
    from art_amd.engine import MipsEngine
    from tests import gpu_util as gu

    rows, queries, ids = _case(120000, 300, seed=503)
    eng = gu.build_engine(rows, ids)
    q = torch.from_numpy(queries).cuda()
    # 300 queries x top-1000: candidate lists of hundreds of MiB
    out = torch.zeros((MipsEngine.packed_words(300, 1000),), dtype=torch.int32, device="cuda")
    with pytest.raises(ArtmipsError) as e:
        eng.search_local_noalloc(0, q, 1000, out)
    assert "ARTMIPS_NOALLOC_SCRATCH_BYTES" in str(e.value)
    torch.cuda.synchronize()
    assert int(out.abs().max().item()) == 0, "a refused call must not have launched anything"
    eng.close()
