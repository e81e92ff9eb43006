"""Compare a (scores fp32, ids) result with what the REFERENCE's own search_mips_index returned for the same
inputs (tests/golden/fx_ref_search.npz, produced by tests/golden/make_golden_ref_search.py).

The reference ranks fp16-rounded scores with torch.topk, whose order among equal values is unspecified
(art_index.py:283,294), so "identical to the reference" means, per query:
  1. the fp16 roundings of our top-k scores are the reference's distances, position by position;
  2. for every score value above the k-th one, the SET of ids at that value equals the reference's set
     (complete tie groups are the same rows, in whatever order);
  3. at the k-th value itself the list cuts a tie group: our ids there are rows that do score that value,
     and among rows of equal fp32 score the lower passage id comes first (BASELINE.json north_star)."""
import numpy as np


def check_against_reference(ref_dist, ref_idx, scores, ids, full_scores, all_ids):
    """ref_dist fp16 [Q,k], ref_idx int32 [Q,k]: the reference's output; scores fp32 [Q,k], ids [Q,k]: ours;
    full_scores float64 [Q,N]: exact scores of every row (oracle), all_ids [N]."""
    ref_dist = np.asarray(ref_dist)
    q, k = ref_dist.shape
    assert scores.shape == (q, k) and ids.shape == (q, k)
    mine16 = np.asarray(scores, dtype=np.float32).astype(np.float16)
    assert np.array_equal(mine16, ref_dist), "fp16(scores) differ from the reference's distances"
    full16 = full_scores.astype(np.float32).astype(np.float16)
    n_groups = 0
    for j in range(q):
        last = ref_dist[j, -1]
        for v in np.unique(ref_dist[j]):
            sel_ref = set(ref_idx[j][ref_dist[j] == v].tolist())
            sel_mine = set(ids[j][mine16[j] == v].tolist())
            if v != last:
                assert sel_ref == sel_mine, "query %d: rows at score %r differ from the reference's" % (j, float(v))
                n_groups += 1
            else:
                tied = set(all_ids[full16[j] == v].tolist())
                assert sel_mine <= tied and sel_ref <= tied and len(sel_mine) == len(sel_ref)
        # our order inside equal fp32 scores: ascending id
        d = np.diff(np.asarray(scores[j], dtype=np.float32))
        assert np.all(d <= 0)
        assert np.all(np.diff(ids[j].astype(np.int64))[d == 0] > 0)
    return n_groups
