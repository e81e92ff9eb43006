"""CPU: the oracle against the outputs of the reference's OWN search_mips_index (fx_ref_search.npz, produced
in the build container by executing /root/reference/megatron/data/art_index.py with device placement
redirected to the CPU: tests/golden/make_golden_ref_search.py).  This is what pins the search oracle:
`exact_topk` reproduces the reference's distances bit for bit and its ids up to torch.topk's unspecified
order inside groups of equal fp16 score, and `reference_restatement` reproduces its output exactly."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import mips_oracle as orc
from tests.ref_search_check import check_against_reference


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "fx_ref_search.npz"))


def _cases(fx, fam):
    for key in fx.files:
        m = re.match(fam + r"_g(\d)_k(\d+)_dist$", key)
        if m:
            yield int(m.group(1)), int(m.group(2)), fx[key], fx[key.replace("_dist", "_idx")]


@pytest.mark.parametrize("fam", ["int", "rnd"])
def test_exact_topk_matches_the_reference_output(fx, fam):
    rows = fx[fam + "_rows"].astype(np.float64)
    queries = fx[fam + "_queries"].astype(np.float64)
    ids = fx[fam + "_ids"]
    seen = groups = 0
    for g, k, ref_dist, ref_idx in _cases(fx, fam):
        s, i, full = orc.exact_topk(queries, rows, ids, k)
        groups += check_against_reference(ref_dist, ref_idx, s, i, full, ids)
        seen += 1
    assert seen >= 4 and groups > 100


def test_reference_output_does_not_depend_on_its_shard_count(fx):
    # art_index.py:251-291: chunk, matmul per chunk, concatenate: the distances are the same for 1, 2, 3 "GPUs"
    for k in (1, 32, 100):
        d1 = fx["int_g1_k%d_dist" % k]
        for g in (2, 3):
            assert np.array_equal(fx["int_g%d_k%d_dist" % (g, k)], d1)
    assert fx["int_g3_chunks"].tolist() == [335, 333] and fx["int_g2_chunks"].tolist() == [502, 501]


@pytest.mark.parametrize("fam", ["int", "rnd"])
def test_restatement_is_the_reference(fx, fam):
    """oracle.reference_restatement (the timed CPU baseline) against the reference's own run on the same
    fp16 tensors: identical distances; identical ids too, because both end in the same torch.topk call on
    the same matrix."""
    rows = torch.from_numpy(fx[fam + "_rows"])
    queries = torch.from_numpy(fx[fam + "_queries"])
    ids = fx[fam + "_ids"]
    for g, k, ref_dist, ref_idx in _cases(fx, fam):
        d, i = orc.reference_restatement(queries, rows, ids, k, n_shards=g)
        assert np.array_equal(d.numpy(), ref_dist)
        assert np.array_equal(i.numpy(), ref_idx)


@pytest.mark.skipif(not os.path.exists("/root/reference/megatron/data/art_index.py"),
                    reason="the reference is only present in the build container")
def test_committed_fixture_is_what_the_reference_produces(fx, golden_dir):
    """Guard of the pin itself: re-run the reference's own add_embed_data + search_mips_index
    (tests/golden/make_golden_ref_search.py, device placement redirected to the CPU) on the inputs stored
    in fx_ref_search.npz and require the stored outputs, for every (family, shards, k) case."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden_ref_search",
                                                  os.path.join(golden_dir, "make_golden_ref_search.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    ref = gen.load_reference()
    cases = 0
    for fam in ("int", "rnd"):
        rows, queries, ids = fx[fam + "_rows"], fx[fam + "_queries"], fx[fam + "_ids"]
        for g, k, want_dist, want_idx in _cases(fx, fam):
            dist, idx, chunks = gen.run_reference(ref, rows, ids, queries, k, g)
            assert np.array_equal(dist.view(np.uint16), want_dist.view(np.uint16)), (fam, g, k)
            assert np.array_equal(idx, want_idx), (fam, g, k)
            key = "%s_g%d_chunks" % (fam, g)
            if key in fx.files:
                assert list(chunks) == fx[key].tolist()
            cases += 1
    assert cases == 13
