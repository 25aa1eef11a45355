"""Oracle: exact squared-L2 kNN and recall@N (TEST INFRASTRUCTURE).

Reference call sites:
    test.py:27-32      faiss.IndexFlatL2(d); .add(db); D, I = .search(q, max(recall_values))
    test.py:73-83      recall@N by set membership against positives_per_query
    datasets/datasets_ws_nuscenes.py:1241-1258   mining: best positive (k=1),
                                                 hardest negatives (k=10 over <=1000 rows)

faiss-cpu is a third-party dependency (unpinned, README.md:48), absent from
/root/reference and from this image -> PARITY UNPINNED; restated from its published
behaviour: IndexFlatL2.search returns SQUARED L2 distances (float32) in ascending
order with int64 labels; slots beyond ntotal hold (FLT_MAX, -1).  For nq >= 20 faiss
evaluates ||x||^2 + ||y||^2 - 2<x,y> with BLAS sgemm (clamped at 0), otherwise the
direct sum of squared differences, so even faiss is only defined up to fp32 rounding;
the judge of record here is the fp64 direct sum.  Tie policy of this oracle: equal
distances are ordered by ascending database index (stable sort).
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def knn_l2_fp64(xq, xb, k):
    """Exact kNN: float64 sum((x-y)^2), stable ascending. Returns (D f32[nq,k], I i64[nq,k])."""
    xq = np.ascontiguousarray(xq, dtype=np.float64)
    xb = np.ascontiguousarray(xb, dtype=np.float64)
    nq, nb = xq.shape[0], xb.shape[0]
    D = np.full((nq, k), FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    D64 = np.full((nq, k), np.inf, dtype=np.float64)
    if nb == 0 or nq == 0:
        return D, I, D64
    bn = (xb * xb).sum(1)
    step = max(1, int(2 ** 26 // max(nb, 1)))
    for s in range(0, nq, step):
        q = xq[s:s + step]
        # direct form in fp64: (q-b)^2 summed; expansion is exact enough in fp64 but we
        # keep the direct sum for small problems and the expansion + refinement for big.
        d = (q * q).sum(1)[:, None] + bn[None, :] - 2.0 * (q @ xb.T)
        kk = min(k, nb)
        # candidate shortlist then exact direct re-evaluation (removes expansion rounding)
        m = min(nb, kk + 32)
        cand = np.argpartition(d, m - 1, axis=1)[:, :m] if m < nb else np.tile(np.arange(nb), (q.shape[0], 1))
        diff = q[:, None, :] - xb[cand]
        dc = (diff * diff).sum(-1)
        order = np.lexsort((cand, dc), axis=1)[:, :kk]
        rows = np.arange(q.shape[0])[:, None]
        I[s:s + step, :kk] = cand[rows, order]
        D64[s:s + step, :kk] = dc[rows, order]
        D[s:s + step, :kk] = dc[rows, order].astype(np.float32)
    return D, I, D64


def knn_l2_faisslike_fp32(xq, xb, k):
    """fp32 restatement of faiss's BLAS path: ||x||^2+||y||^2-2<x,y>, clamp 0, top-k selection (argpartition to the k
    smallest, then a stable sort of those k by (distance, index)) -- sgemm + selection, not a full sort of the row."""
    xq = np.ascontiguousarray(xq, dtype=np.float32)
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    nq, nb = xq.shape[0], xb.shape[0]
    D = np.full((nq, k), FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if nb == 0 or nq == 0:
        return D, I
    bn = (xb * xb).sum(1, dtype=np.float32)
    qn = (xq * xq).sum(1, dtype=np.float32)
    kk = min(k, nb)
    step = max(1, int(2 ** 26 // max(nb, 1)))
    for s in range(0, nq, step):
        d = qn[s:s + step, None] + bn[None, :] - np.float32(2.0) * (xq[s:s + step] @ xb.T)
        np.maximum(d, 0, out=d)
        rows = np.arange(d.shape[0])[:, None]
        if kk < nb:
            part = np.argpartition(d, kk - 1, axis=1)[:, :kk]
            # ties AT the k-th distance: argpartition may keep any of the equal rows; take every row <= the k-th value
            # whenever a tie straddles the cut so that the (distance, index) order decides, as a stable full sort would
            kth = d[rows, part].max(axis=1)
            tie = (d <= kth[:, None]).sum(axis=1) > kk
            dk = d[rows, part]
            order = np.lexsort((part, dk), axis=1)
            idx = part[rows, order]
            for r in np.nonzero(tie)[0]:
                c = np.nonzero(d[r] <= kth[r])[0]
                idx[r] = c[np.lexsort((c, d[r, c]))][:kk]
        else:
            idx = np.argsort(d, axis=1, kind="stable")[:, :kk]
        I[s:s + step, :kk] = idx
        D[s:s + step, :kk] = d[rows, idx]
    return D, I


def recall_at(predictions, positives_per_query, recall_values):
    """reference test.py:73-83: recalls in percent, cumulative over recall_values."""
    recalls = np.zeros(len(recall_values))
    for qi, pred in enumerate(predictions):
        for i, n in enumerate(recall_values):
            if np.any(np.isin(pred[:n], positives_per_query[qi])):
                recalls[i:] += 1
                break
    return recalls / len(predictions) * 100


def compute_recall(queries_features, database_features, positives_per_query,
                   recall_values=(1, 5, 10, 20)):
    """reference test.py:24-84 (compute_recall, test_method='hard_resize')."""
    _, predictions, _ = knn_l2_fp64(queries_features, database_features, max(recall_values))
    recalls = recall_at(predictions, positives_per_query, list(recall_values))
    recalls_str = ", ".join(f"R@{v}: {r:.1f}" for v, r in zip(recall_values, recalls))
    return recalls, recalls_str


def unambiguous_mask(D64, rel_gap=1e-6):
    """True where the fp64 gap to BOTH neighbours in the sorted list exceeds rel_gap
    (parity = same index wherever the gap > 1e-6 relative)."""
    d = D64
    gap_next = np.abs(np.diff(d, axis=1))
    scale = np.maximum(np.abs(d[:, :-1]), 1e-30)
    ok_pair = gap_next > rel_gap * scale
    ok = np.ones(d.shape, dtype=bool)
    ok[:, :-1] &= ok_pair
    ok[:, 1:] &= ok_pair
    return ok
