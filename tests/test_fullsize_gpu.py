"""Full-size (21,015,324 x 768 fp16, BASELINE config 2) checks through size-independent properties:
planted winners at known rows must come back in the planted order, the rest of the list must agree
with an independent chunked torch reference, and a 4-shard split of the same rows must return the
same ids."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 21015324
DIM = 768


@pytest.fixture(scope="module")
def big():
    from art_amd.engine import MipsEngine
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    queries = torch.randn((64, DIM), generator=g, device=dev).half()
    # planted rows: first, last, around the 4- and 8-way shard boundaries, inside the last partial tile
    chunk8 = -(-N // 8)
    planted = [0, N - 1, N - 3, chunk8 - 1, chunk8, 2 * chunk8 - 1, 4 * chunk8, 7 * chunk8, 12345677]
    eng = MipsEngine(DIM, torch.float16, [0])
    eng4 = MipsEngine(DIM, torch.float16, [0, 0, 0, 0])
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    chunk4 = -(-N // 4)
    eng.begin(N)
    eng4.begin(N)
    step = 1 << 18
    best_s = torch.full((64, 40), -float("inf"), device=dev)
    best_i = torch.zeros((64, 40), dtype=torch.int64, device=dev)
    for off in range(0, N, step):
        n = min(step, N - off)
        rows = (torch.randn((n, DIM), generator=gen, device=dev) * (1.0 / math.sqrt(DIM))).half()
        for j, p in enumerate(planted):
            if off <= p < off + n:
                # query j's own direction, scaled so that the score (~768 * scale) beats every random row
                rows[p - off] = (queries[j].float() * (0.02 + 0.001 * j)).half()
        ids = torch.arange(off + 1, off + n + 1, device=dev, dtype=torch.int32)
        eng.upload_device(0, rows, ids, off)
        # the same rows into the 4-shard engine (a chunk may straddle a shard boundary)
        pos = off
        while pos < off + n:
            slot = pos // chunk4
            end = min(off + n, (slot + 1) * chunk4)
            eng4.upload_device(slot, rows[pos - off:end - off], ids[pos - off:end - off], pos - slot * chunk4)
            pos = end
        # independent reference: rocBLAS matmul (fp32 accumulate) + running torch.topk
        s = rows.float() @ queries.float().T
        cs, ci = torch.topk(s.T, 40, dim=1)
        cat_s = torch.cat([best_s, cs], 1)
        cat_i = torch.cat([best_i, ci + off + 1], 1)
        o = torch.argsort(cat_s, dim=1, descending=True, stable=True)[:, :40]
        best_s, best_i = torch.gather(cat_s, 1, o), torch.gather(cat_i, 1, o)
        del rows, s
    eng.finalize()
    eng4.finalize()
    yield eng, eng4, queries, planted, best_s, best_i
    eng.close()
    eng4.close()


def test_planted_winners_and_reference(big):
    eng, eng4, queries, planted, ref_s, ref_i = big
    s, i = eng.search(queries, 32, score_dtype=torch.float32)
    assert eng.info()["last_retries"] == 0
    for j, p in enumerate(planted):
        assert int(i[j, 0]) == p + 1, "planted row %d not first for query %d" % (p, j)
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    # against the torch reference: same ids wherever neighbouring reference scores are > 1e-5 apart
    gap = (ref_s[:, :33][:, :-1] - ref_s[:, :33][:, 1:])
    clear = (gap[:, :32] > 1e-4) & (torch.cat([torch.ones_like(gap[:, :1]), gap[:, :31]], 1) > 1e-4)
    assert clear.float().mean() > 0.9
    assert bool((i.long()[clear] == ref_i[:, :32][clear]).all())
    assert torch.allclose(s, ref_s[:, :32], rtol=1e-3, atol=1e-4)


def test_four_shards_equal_one(big):
    eng, eng4, queries, planted, ref_s, ref_i = big
    s1, i1 = eng.search(queries, 32, score_dtype=torch.float32)
    s4, i4 = eng4.search(queries, 32, score_dtype=torch.float32)
    assert eng4.info()["rows_per_shard"] == [5253831, 5253831, 5253831, 5253831]
    assert torch.equal(i1, i4) and torch.equal(s1, s4)


def test_eval_shape_k100_q16(big):
    # evaluate.py's call shape: 16 queries, top-100
    eng, eng4, queries, planted, ref_s, ref_i = big
    s, i = eng.search(queries[:16], 100, score_dtype=torch.float32)
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    assert bool((i.long()[:, :32] == ref_i[:16, :32]).float().mean() > 0.95)
    for j in range(9):
        assert int(i[j, 0]) == planted[j] + 1


def test_large_batch_shape_equals_training_shape(big):
    # 320 queries in one call (8 waves x 2 groups of 128 per sweep + a 64-query remainder) must give,
    # query by query, what the 64-query training shape gives: same MFMA k order, so bit-equal scores
    eng, eng4, queries, planted, ref_s, ref_i = big
    s64, i64 = eng.search(queries, 32, score_dtype=torch.float32)
    perm = torch.randperm(320, generator=torch.Generator().manual_seed(7)).to(queries.device)
    q320 = queries.repeat(5, 1)[perm]
    s, i = eng.search(q320, 32, score_dtype=torch.float32)
    assert eng.info()["last_retries"] == 0 and eng.info()["last_scan_launches"] == 2
    src = perm % 64
    assert torch.equal(i, i64[src]) and torch.equal(s, s64[src])
    # and through four shards + merge
    s4, i4 = eng4.search(q320, 32, score_dtype=torch.float32)
    assert torch.equal(i4, i) and torch.equal(s4, s)


def test_c3_eval_set_one_call(big):
    """BASELINE config 3 on one GPU and on four shards of it: 3,610 queries, top-100, one call.  The first
    64 queries are the fixture's (planted winners, reference lists); batch position must not matter
    (bit-identical to the 64-query call) and 4 shards + merge must equal one shard everywhere."""
    eng, eng4, queries, planted, ref_s, ref_i = big
    g = torch.Generator(device=queries.device)
    g.manual_seed(99)
    more = torch.randn((3610 - 64, DIM), generator=g, device=queries.device).half()
    q = torch.cat([queries, more], 0)
    s, i = eng.search(q, 100, score_dtype=torch.float32)
    info = eng.info()
    # on a 256-CU part: gangs of 8, 4 and 2 workgroups x 256 queries (2,048 + 1,024 + 512), and the last 26
    # queries on their own; a part whose CU count allows smaller gangs only takes more launches, never more
    # than one per 256 queries
    assert info["last_retries"] == 0
    if info["n_cu"] % 64 == 0:
        assert info["last_scan_launches"] == 4
    assert 4 <= info["last_scan_launches"] <= -(-3610 // 256)
    assert bool((s[:, :-1] >= s[:, 1:]).all()) and bool((i > 0).all()) and bool((i <= N).all())
    for j, p in enumerate(planted):
        assert int(i[j, 0]) == p + 1
    s64, i64 = eng.search(queries, 100, score_dtype=torch.float32)
    assert torch.equal(i[:64], i64) and torch.equal(s[:64], s64)
    assert bool((i.long()[:64, :32] == ref_i[:, :32]).float().mean() > 0.95)
    s4, i4 = eng4.search(q, 100, score_dtype=torch.float32)
    assert torch.equal(i4, i) and torch.equal(s4, s)
    # the reference's own chunking (--shard-size 16, evaluate.py:149-160) on a slice of the set
    outs = [eng.search(q[a:a + 16], 100, score_dtype=torch.float32) for a in range(3600 - 64, 3610, 16)]
    assert torch.equal(torch.cat([o[1] for o in outs]), i[3600 - 64:])
