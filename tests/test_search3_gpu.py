"""GPU tests added in round 3: inputs the C ABI does not rule out (repeated passage ids, a merge fed the
same list twice) must still fill every output slot; the other-device branch of the one-process /
many-shards search (peer copy of the queries, private list, peer copy of the list:
/root/reference/megatron/data/art_index.py:273-291 is the reference's form of it) executed on one GPU
through the diagnostic build's ARTMIPS_FORCE_REMOTE."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mips_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gu():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from tests import gpu_util

    return gpu_util


def _search_prefilled(eng, queries, k):
    """artmips_search through the C ABI into outputs that start as (NaN, -7): a slot the kernels never
    write keeps that value (torch.empty would usually hand back zeros and hide it)."""
    from art_amd import _lib

    q = torch.from_numpy(queries).cuda()
    scores = torch.full((q.shape[0], k), float("nan"), dtype=torch.float32, device="cuda")
    ids = torch.full((q.shape[0], k), -7, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.artmips_search(eng._h, ctypes.c_void_p(q.data_ptr()), _lib.F16, q.shape[0], k,
                                       ctypes.c_void_p(scores.data_ptr()), _lib.F32,
                                       ctypes.c_void_p(ids.data_ptr()),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return scores.cpu().numpy(), ids.cpu().numpy()


@pytest.mark.parametrize("n,k", [(3000, 32), (60000, 32), (60000, 100)])
def test_repeated_ids_leave_no_unwritten_slot(gu, n, k):
    """Identical rows under identical ids (artmips_upload does not look for repeated ids): equal
    64-bit keys.  Every output slot must be written and the (score, id) multiset must be the oracle's."""
    rng = np.random.Generator(np.random.PCG64(7000 + n + k))
    rows = rng.integers(-3, 4, size=(n, 768)).astype(np.float16)   # integer scores: exact in any order
    queries = rng.integers(-2, 3, size=(5, 768)).astype(np.float16)
    ids = (rng.permutation(n) + 1).astype(np.int32)
    # plant copies of the best rows of query 0 under their own ids: equal keys inside the top-k
    best = np.argsort(-(rows.astype(np.float64) @ queries[0].astype(np.float64)), kind="stable")[:6]
    for t, src in enumerate(best):
        dst = (int(src) + 1000 + 17 * t) % n
        rows[dst] = rows[src]
        ids[dst] = ids[src]
    eng = gu.build_engine(rows, ids)
    s, i = _search_prefilled(eng, queries, k)
    eng.close()
    assert not np.isnan(s).any() and (i != -7).all(), "an output slot was never written"
    full = queries.astype(np.float64) @ rows.astype(np.float64).T
    for qi in range(queries.shape[0]):
        order = np.lexsort((ids, -full[qi]))[:k]
        want = sorted(zip(full[qi][order].tolist(), ids[order].tolist()), key=lambda p: (-p[0], p[1]))
        got = sorted(zip(s[qi].astype(np.float64).tolist(), i[qi].tolist()), key=lambda p: (-p[0], p[1]))
        # rows tied with the k-th score and id may be exchanged for one another only if they are equal keys
        assert got == want, "query %d" % qi
        assert (np.diff(s[qi]) <= 0).all()


@pytest.mark.parametrize("k,lists", [(32, 2), (100, 3), (1000, 2)])
def test_merge_of_repeated_lists_fills_every_slot(gu, k, lists):
    """artmips_merge fed the same sorted list several times: every key occurs `lists` times."""
    from art_amd import _lib

    rng = np.random.Generator(np.random.PCG64(7100 + k))
    nq = 4
    sc = np.sort(rng.standard_normal((nq, k)).astype(np.float32), axis=1)[:, ::-1].copy()
    idv = rng.permutation(10 * k)[:nq * k].reshape(nq, k).astype(np.int32) + 1
    scores = torch.from_numpy(np.stack([sc] * lists)).cuda().contiguous()
    ids = torch.from_numpy(np.stack([idv] * lists)).cuda().contiguous()
    out_s = torch.full((nq, k), float("nan"), dtype=torch.float32, device="cuda")
    out_i = torch.full((nq, k), -7, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.artmips_merge(0, ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                                      lists, nq, k, ctypes.c_void_p(out_s.data_ptr()), _lib.F32,
                                      ctypes.c_void_p(out_i.data_ptr()),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got_s, got_i = out_s.cpu().numpy(), out_i.cpu().numpy()
    assert not np.isnan(got_s).any() and (got_i != -7).all(), "an output slot was never written"
    for qi in range(nq):
        want_s = np.repeat(sc[qi], lists)[:k]
        want_i = np.repeat(idv[qi], lists)[:k]
        assert np.array_equal(got_s[qi], want_s) and np.array_equal(got_i[qi], want_i), "query %d" % qi


def test_remote_branch_of_the_one_process_search_on_one_gpu():
    """multi_enqueue's `remote` branch (artmips.hip) runs between two real devices only; the diagnostic
    build forces it for shards on slot 0's own device (ARTMIPS_FORCE_REMOTE, read on every call), in a
    process of its own: 4 shards, 64 x top-32 and 600 x top-1000, bit-identical to the local branch and
    right by the oracle; then the adversarial input whose candidate lists overflow (retry through the
    remote branch)."""
    code = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from art_amd import _lib
assert _lib.LIB_PATH.endswith("libartmips_diag.so")
from tests import gpu_util as gu
from oracle import mips_oracle as orc

def both(eng, q, k):
    os.environ["ARTMIPS_FORCE_REMOTE"] = "0"
    s0, i0 = eng.search(q, k, score_dtype=torch.float32)
    packed_before = eng.info()["workspace_bytes"]
    os.environ["ARTMIPS_FORCE_REMOTE"] = "1"
    s1, i1 = eng.search(q, k, score_dtype=torch.float32)
    retries = eng.info()["last_retries"]
    # the remote branch owns per-shard query copies and private lists: the workspace must have grown
    assert eng.info()["workspace_bytes"] > packed_before, "the remote branch did not run"
    s2, i2, st = eng.search_async(q, k, score_dtype=torch.float32)
    torch.cuda.synchronize()
    assert int(st[0].item()) == 0
    os.environ["ARTMIPS_FORCE_REMOTE"] = "0"
    assert torch.equal(i0, i1) and torch.equal(s0, s1), "remote branch differs from the local branch"
    assert torch.equal(i0, i2) and torch.equal(s0, s2), "async remote branch differs"
    return s1.cpu().numpy(), i1.cpu().numpy(), retries

rng = np.random.Generator(np.random.PCG64(7200))
n = 90000
rows = (rng.standard_normal((n, 768)) * 0.05).astype(np.float16)
ids = (rng.permutation(n) + 1).astype(np.int32)
eng = gu.build_engine(rows, ids, devices=[0, 0, 0, 0])
for nq, k in ((64, 32), (600, 1000)):
    queries = rng.standard_normal((nq, 768)).astype(np.float16)
    s, i, _ = both(eng, torch.from_numpy(queries).cuda(), k)
    orc.check_topk(s, i, queries.astype(np.float64), rows.astype(np.float64), ids, k, rel_tol=1e-3)
eng.close()

# overflow retry through the remote branch (the input of test_multi_shard_overflow_retry_is_exact)
n, k = 262144, 8
rng = np.random.Generator(np.random.PCG64(12))
rows = (rng.standard_normal((n, 768)) * 0.05).astype(np.float16)
queries = rng.standard_normal((3, 768)).astype(np.float16)
ids = np.arange(1, n + 1, dtype=np.int32)
tiles = np.arange(n // 16)
for t in tiles[tiles %% 8 == 0]:
    rows[t * 16:(t + 1) * 16] = 0
eng = gu.build_engine(rows, ids, devices=[0, 0])
os.environ["ARTMIPS_FORCE_REMOTE"] = "1"
s, i = eng.search(torch.from_numpy(queries).cuda(), k, score_dtype=torch.float32)
assert eng.info()["last_retries"] >= 1, "this input is meant to overflow the candidate lists"
orc.check_topk(s.cpu().numpy(), i.cpu().numpy(), queries.astype(np.float64), rows.astype(np.float64), ids, k,
               rel_tol=1e-3)
eng.close()
print("remote ok")
''' % ROOT
    env = dict(os.environ)
    env["ARTMIPS_DIAGNOSTICS"] = "1"
    env["ARTMIPS_LIBRARY"] = os.path.join(ROOT, "art_amd", "lib", "libartmips_diag.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "remote ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_gang_rendezvous_changes_timing_only():
    """The rendezvous between the members of a gang (scan_kernel GANG_SYNC) only ever pauses a workgroup: the
    diagnostic build switches it off (ARTMIPS_ABLATE 8192) and makes it pause all the time (a lead of 1 stage);
    ids and scores must be bit-identical in all three, in both gang shapes (4 x 4 at top-100, 8 x 2 at top-1000),
    through the two-part scan, and right by the oracle."""
    code = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from art_amd import _lib
assert _lib.LIB_PATH.endswith("libartmips_diag.so")
from tests import gpu_util as gu
from oracle import mips_oracle as orc

rng = np.random.Generator(np.random.PCG64(8100))
n = 140000   # enough tiles for gangs of 8 and for the two-part scan
rows = (rng.standard_normal((n, 768)) * 0.05).astype(np.float16)
ids = (rng.permutation(n) + 1).astype(np.int32)
eng = gu.build_engine(rows, ids)
for nq, k in ((2048, 100), (2048, 1000), (1024, 300)):
    queries = rng.standard_normal((nq, 768)).astype(np.float16)
    q = torch.from_numpy(queries).cuda()
    out = []
    for ablate, lead in (("0", "0"), ("8192", "0"), ("0", "1")):
        os.environ["ARTMIPS_ABLATE"] = ablate
        os.environ["ARTMIPS_SYNC_LEAD"] = lead
        s, i = eng.search(q, k, score_dtype=torch.float32)
        assert eng.info()["last_retries"] == 0
        out.append((s, i))
    os.environ["ARTMIPS_ABLATE"] = "0"
    os.environ["ARTMIPS_SYNC_LEAD"] = "0"
    for s, i in out[1:]:
        assert torch.equal(i, out[0][1]) and torch.equal(s, out[0][0]), "the rendezvous changed a result"
    n_diff = orc.check_topk(out[0][0].cpu().numpy(), out[0][1].cpu().numpy(), queries.astype(np.float64),
                            rows.astype(np.float64), ids, k, rel_tol=1e-3)
    assert n_diff <= max(2, (nq * k) // 500)
eng.close()
print("rendezvous ok")
''' % ROOT
    env = dict(os.environ)
    env["ARTMIPS_DIAGNOSTICS"] = "1"
    env["ARTMIPS_LIBRARY"] = os.path.join(ROOT, "art_amd", "lib", "libartmips_diag.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and "rendezvous ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_stray_library_variable_is_ignored_without_the_opt_in():
    """`import art_amd` loads the shipped library unless BOTH ARTMIPS_DIAGNOSTICS=1 and ARTMIPS_LIBRARY
    are set (art_amd/_lib.py)."""
    code = ("import sys, warnings; sys.path.insert(0, %r)\n"
            "warnings.simplefilter('ignore')\n"
            "from art_amd import _lib\n"
            "print(_lib.LIB_PATH)\n" % ROOT)
    env = dict(os.environ)
    env.pop("ARTMIPS_DIAGNOSTICS", None)
    env["ARTMIPS_LIBRARY"] = os.path.join(ROOT, "art_amd", "lib", "libartmips_diag.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("libartmips.so"), r.stdout + r.stderr


# ---- every multiple of 128 up to 1,024 (round 3): the widths between the three the first rounds built
@pytest.mark.parametrize("dim", [256, 384, 512, 640, 896])
@pytest.mark.parametrize("storage", ["f16", "bf16"])
def test_other_widths_vs_oracle(gu, dim, storage):
    st = torch.float16 if storage == "f16" else torch.bfloat16
    rng = np.random.Generator(np.random.PCG64(7300 + dim))
    for n, q, k in ((3001, 5, 17), (40000, 150, 32), (30000, 9, 300)):   # no pass 0 | pass 0 + three sweeps | large k
        rows = (rng.standard_normal((n, dim)) * 0.05).astype(np.float32)
        queries = rng.standard_normal((q, dim)).astype(np.float32)
        ids = (rng.permutation(n) + 1).astype(np.int32)
        eng = gu.build_engine(rows, ids, storage=st)
        s, i = gu.search_np(eng, queries, k, q_dtype=torch.float32)
        n_diff = orc.check_topk(s, i, orc.to_storage(queries, storage), orc.to_storage(rows, storage), ids, k,
                                rel_tol=1e-3)
        assert n_diff <= max(2, (q * k) // 500), "%d id positions differ from the oracle" % n_diff
        assert eng.info()["last_retries"] == 0
        eng.close()


@pytest.mark.parametrize("dim", [256, 640, 896])
def test_other_widths_integer_ties_bit_exact(gu, dim):
    """Integer-valued rows with planted duplicates: every dot product is exact in fp32 in any order, ties
    are real, and the answer must equal the oracle's bit for bit (score desc, id asc)."""
    rng = np.random.Generator(np.random.PCG64(7400 + dim))
    n, q, k = 20000, 6, 40
    rows = rng.integers(-4, 5, size=(n, dim)).astype(np.float16)
    queries = rng.integers(-4, 5, size=(q, dim)).astype(np.float16)
    for t in range(30):
        rows[(997 * t + 5) % n] = rows[t]          # duplicates of rows that may well be in a top-k
    ids = (rng.permutation(n) + 1).astype(np.int32)
    eng = gu.build_engine(rows, ids)
    s, i = gu.search_np(eng, queries, k)
    want_s, want_i, _ = orc.exact_topk(queries.astype(np.float64), rows.astype(np.float64), ids, k)
    assert np.array_equal(i, want_i) and np.array_equal(s, want_s)
    eng.close()
