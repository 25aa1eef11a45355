"""Batched triplet mining on the GPU.

The reference mines per query in a python loop (`compute_triplets_partial_sep`,
datasets/datasets_ws_nuscenes.py:1372-1410): for each of `cache_refresh_rate` sampled queries it
builds TWO faiss indexes (best positive :1241-1248, hardest negatives :1250-1258) from rows of a
host-side feature cache.  Here the cache stays on the device and the whole refresh is
    * one `agp_mine_best_positive` launch over the ragged positive lists, and
    * one `agp_knn_search` over the sampled database rows with k = negs + (most soft positives any
      query has inside the sample), followed by dropping each query's soft positives;
results are identical to the per-query loop: same candidate order (np.setdiff1d sorts), same tie rule
(earlier candidate wins), exact distances.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from .retrieval import IndexFlatL2


def _dev(x, device, dtype):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=dtype).contiguous()


def _csr(lists, device):
    lens = [len(x) for x in lists]
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.concatenate([np.asarray(x, dtype=np.int64).reshape(-1) for x in lists]) if off[-1] else \
        np.zeros(0, dtype=np.int64)
    return torch.from_numpy(off).to(device), torch.from_numpy(flat).to(device)


def best_positive_indexes(query_features, database_features, hard_positives_per_query, device="cuda"):
    """[Q] int64: for query i the row of `database_features` among hard_positives_per_query[i] nearest
    in feature space (reference get_best_positive_index, :1241-1248); -1 for an empty list."""
    dev = torch.device(device)
    xq = _dev(query_features, dev, torch.float32)
    xb = _dev(database_features, dev, torch.float32)
    off, idx = _csr(hard_positives_per_query, dev)
    if idx.numel() == 0:
        idx = torch.zeros(1, dtype=torch.int64, device=dev)
    out = torch.empty(xq.shape[0], dtype=torch.int64, device=dev)
    check(_lib.load().agp_mine_best_positive(ptr(xq), xq.shape[0], ptr(xb), xb.shape[0], xq.shape[1], ptr(off), ptr(idx),
                                             ptr(out), None, _lib.stream()), "agp_mine_best_positive")
    return out


def hardest_negatives_indexes(query_features, database_features, sampled_database_indexes, soft_positives_per_query,
                              negs_num_per_query=10, device="cuda"):
    """[Q, negs] int64 database rows: per query the `negs` nearest rows of
    setdiff1d(sampled_database_indexes, soft_positives) (reference :1394-1404 + :1250-1258)."""
    dev = torch.device(device)
    xq = _dev(query_features, dev, torch.float32)
    xb = _dev(database_features, dev, torch.float32)
    cand = np.unique(np.asarray(sampled_database_indexes, dtype=np.int64))       # setdiff1d's order: sorted
    nq, nc = xq.shape[0], cand.shape[0]
    # soft positives that actually lie in the sample, as (query, candidate position) pairs
    pos_q, pos_c = [], []
    most = 0
    for qi, soft in enumerate(soft_positives_per_query):
        soft = np.asarray(soft, dtype=np.int64).reshape(-1)
        if soft.size == 0:
            continue
        where = np.searchsorted(cand, soft)
        hit = (where < nc) & (cand[np.minimum(where, nc - 1)] == soft)
        hits = np.unique(where[hit])
        most = max(most, hits.size)
        pos_q.append(np.full(hits.size, qi, dtype=np.int64))
        pos_c.append(hits)
    k = min(nc, negs_num_per_query + most)
    if nc - most < negs_num_per_query:
        raise ValueError("fewer candidate negatives than negs_num_per_query for some query")
    cand_t = torch.from_numpy(cand).to(dev)
    index = IndexFlatL2(xq.shape[1], device=dev)
    index.add(xb.index_select(0, cand_t))
    _, I = index.search_device(xq, k)                                          # [Q,k] positions in cand, nearest first
    if most:
        keys = torch.from_numpy(np.concatenate(pos_q) * nc + np.concatenate(pos_c)).to(dev)
        rowkey = torch.arange(nq, device=dev).view(-1, 1) * nc + I
        keep = ~torch.isin(rowkey, keys)
    else:
        keep = torch.ones_like(I, dtype=torch.bool)
    rank = torch.cumsum(keep, 1) - 1                                            # rank among kept entries
    sel = keep & (rank < negs_num_per_query)
    out = torch.empty((nq, negs_num_per_query), dtype=torch.int64, device=dev)
    rows = torch.arange(nq, device=dev).view(-1, 1).expand_as(I)
    out[rows[sel], rank[sel]] = cand_t[I[sel]]
    return out


def compute_triplets_partial(query_features, database_features, sampled_queries_indexes, hard_positives_per_query,
                             soft_positives_per_query, sampled_database_indexes, negs_num_per_query=10, device="cuda"):
    """The triplet table of `compute_triplets_partial_sep` (:1372-1410): int64 [Q, 2 + negs] rows
    (query_index, best_positive_index, neg_indexes...).  `query_features[i]` belongs to
    sampled_queries_indexes[i]; the two per-query lists are indexed by the GLOBAL query index, as
    in the reference dataset object."""
    sq = np.asarray(sampled_queries_indexes, dtype=np.int64)
    hard = [hard_positives_per_query[i] for i in sq]
    soft = [soft_positives_per_query[i] for i in sq]
    best = best_positive_indexes(query_features, database_features, hard, device)
    negs = hardest_negatives_indexes(query_features, database_features, sampled_database_indexes, soft,
                                     negs_num_per_query, device)
    return torch.cat([torch.from_numpy(sq).to(best.device).view(-1, 1), best.view(-1, 1), negs], 1)
