"""-m gpu parity of the batched triplet mining against the per-query oracle loop."""
import numpy as np
import pytest
import torch

from oracle import mining as omining

pytestmark = pytest.mark.gpu


def _case(seed, nq, ndb, d=256, nsample=300, dup=True):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((24, d)).astype(np.float32)
    db = centers[rng.integers(0, 24, ndb)] + 0.3 * rng.standard_normal((ndb, d)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    if dup:
        db[ndb // 2] = db[ndb // 3]                    # exact duplicates: tie rule is exercised
        db[ndb // 5] = db[ndb // 7]
    q = db[rng.integers(0, ndb, nq)] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    hard = [rng.choice(ndb, size=rng.integers(1, 12), replace=False) for _ in range(nq)]
    soft = [np.unique(np.concatenate([h, rng.choice(ndb, size=rng.integers(0, 40), replace=False)])) for h in hard]
    sampled = rng.choice(ndb, size=nsample, replace=False)
    return q, db, hard, soft, sampled


@pytest.mark.parametrize("seed,nq,ndb,nsample", [(0, 37, 500, 200), (1, 260, 3000, 1000), (2, 5, 64, 40)])
def test_triplets_match_per_query_loop(dev, seed, nq, ndb, nsample):
    from agplace_amd import mining
    q, db, hard, soft, sampled = _case(seed, nq, ndb, nsample=nsample)
    qidx = np.arange(nq)
    got = mining.compute_triplets_partial(q, db, qidx, hard, soft, sampled, 10, device=dev).cpu().numpy()
    ref = omining.compute_triplets_partial(q, db, qidx, hard, soft, sampled, 10)
    assert got.shape == (nq, 12)
    assert np.array_equal(got, ref)


def test_best_positive_empty_list_and_device_inputs(dev):
    from agplace_amd import mining
    q, db, hard, soft, sampled = _case(3, 8, 200, nsample=100, dup=False)
    hard[3] = np.zeros(0, dtype=np.int64)
    best = mining.best_positive_indexes(torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev), hard, device=dev).cpu()
    assert int(best[3]) == -1
    for i in (0, 1, 2, 4, 5, 6, 7):
        assert int(best[i]) == omining.best_positive_index(q[i], db, hard[i])


def test_too_few_negatives_is_an_error(dev):
    from agplace_amd import mining
    q, db, hard, soft, sampled = _case(4, 3, 50, nsample=12, dup=False)
    soft[0] = np.asarray(sampled[:6])
    with pytest.raises(ValueError):
        mining.hardest_negatives_indexes(q, db, sampled, soft, 10, device=dev)


def test_many_in_sample_soft_positives_and_sample_order_ties(dev):
    """(a) A query with more in-sample soft positives than agp_knn_search's k limit allows (k = negs + soft positives
    > 128) is searched over its own candidate set instead of aborting the refresh; (b) equal distances resolve to the
    earlier row of the RANDOM sample (np.setdiff1d(..., assume_unique=True) keeps the sample's order), not to the
    smaller database index."""
    from agplace_amd import mining
    q, db, hard, soft, sampled = _case(6, 12, 900, nsample=400, dup=False)
    soft[2] = np.unique(np.concatenate([soft[2], sampled[:150]]))          # 150 in-sample soft positives
    soft[7] = np.unique(np.concatenate([soft[7], sampled[100:330]]))       # 230
    # two sampled rows identical to query 4's nearest row, the larger database index first in the sample
    a, b = int(max(sampled[10], sampled[11])), int(min(sampled[10], sampled[11]))
    sampled = sampled.copy()
    sampled[10], sampled[11] = a, b
    db[a] = q[4]
    db[b] = q[4]
    soft[4] = np.zeros(0, dtype=np.int64)
    qidx = np.arange(12)
    got = mining.compute_triplets_partial(q, db, qidx, hard, soft, sampled, 10, device=dev).cpu().numpy()
    ref = omining.compute_triplets_partial(q, db, qidx, hard, soft, sampled, 10)
    assert np.array_equal(got, ref)
    assert got[4, 2] == a and got[4, 3] == b


def test_triplets_equal_the_references_own_methods(dev, golden):
    """The batched HIP mining against the triplet table of the reference's own get_best_positive_index /
    get_hardest_negatives_indexes (tests/golden/mining.npz, make_golden.py section 10)."""
    from agplace_amd import mining
    g = golden("mining")
    hard = np.split(g["hard_flat"], np.cumsum(g["hard_len"])[:-1])
    soft = np.split(g["soft_flat"], np.cumsum(g["soft_len"])[:-1])
    ndb = int(g["ndb"])
    cache = g["cache"]
    qf = cache[ndb + g["sampled_q"]]
    got = mining.compute_triplets_partial(qf, cache, g["sampled_q"], hard, soft, g["sampled_db"], int(g["negs"]), device=dev).cpu().numpy()
    assert np.array_equal(got, g["triplets"])
