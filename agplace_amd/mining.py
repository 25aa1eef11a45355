"""Batched triplet mining on the GPU.

The reference mines per query in a python loop (`compute_triplets_partial_sep`,
datasets/datasets_ws_nuscenes.py:1372-1410): for each of `cache_refresh_rate` sampled queries it
builds TWO faiss indexes (best positive :1241-1248, hardest negatives :1250-1258) from rows of a
host-side feature cache.  Here the cache stays on the device and the whole refresh is
    * one `agp_mine_best_positive` launch over the ragged positive lists, and
    * one `agp_knn_search` over the sampled database rows with k = negs + (most soft positives any
      query has inside the sample), followed by dropping each query's soft positives;
results are identical to the per-query loop: same candidate order (np.setdiff1d(..., assume_unique=True) keeps the
random sample's order), same tie rule (earlier candidate wins), exact distances.
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


MAX_K = 128      # agp_knn_search's limit on k (csrc/knn.hip)


def hardest_negatives_indexes(query_features, database_features, sampled_database_indexes, soft_positives_per_query,
                              negs_num_per_query=10, device="cuda"):
    """[Q, negs] int64 database rows: per query the `negs` nearest rows of
    np.setdiff1d(sampled_database_indexes, soft_positives, assume_unique=True) (reference :1394-1404 + :1250-1258).
    Candidates keep the order of `sampled_database_indexes` (setdiff1d with assume_unique=True does not sort), so
    equal distances resolve to the earlier SAMPLED row, as an index built over that array would.
    One batched search with k = negs + (most in-sample soft positives of any query) serves every query whose soft
    positives leave k <= 128; the few queries with more in-sample soft positives than that (small or dense
    databases) are searched one by one over their own candidate set, like the reference does for every query."""
    dev = torch.device(device)
    xq = _dev(query_features, dev, torch.float32)
    xb = _dev(database_features, dev, torch.float32)
    cand = np.asarray(sampled_database_indexes, dtype=np.int64).reshape(-1)      # the sample's own order
    nq, nc = xq.shape[0], cand.shape[0]
    order = np.argsort(cand, kind="stable")
    sorted_c = cand[order]
    # soft positives that actually lie in the sample, as candidate POSITIONS per query
    hits_per_q = []
    for soft in soft_positives_per_query:
        soft = np.unique(np.asarray(soft, dtype=np.int64).reshape(-1))
        if soft.size == 0 or nc == 0:
            hits_per_q.append(np.zeros(0, dtype=np.int64))
            continue
        lo, hi = np.searchsorted(sorted_c, soft, "left"), np.searchsorted(sorted_c, soft, "right")
        if np.all(hi - lo <= 1):
            hits = order[lo[hi > lo]]
        else:                                   # a sampled row listed twice: every copy is a soft positive
            hits = np.concatenate([order[a:b] for a, b in zip(lo, hi)]) if lo.size else np.zeros(0, dtype=np.int64)
        hits_per_q.append(np.sort(hits))
    counts = np.array([h.size for h in hits_per_q], dtype=np.int64)
    if nq and nc - int(counts.max(initial=0)) < negs_num_per_query:
        raise ValueError("fewer candidate negatives than negs_num_per_query for some query")
    out = torch.empty((nq, negs_num_per_query), dtype=torch.int64, device=dev)
    cand_t = torch.from_numpy(cand).to(dev)
    batched = np.nonzero(counts <= MAX_K - negs_num_per_query)[0]
    single = np.nonzero(counts > MAX_K - negs_num_per_query)[0]
    if batched.size:
        most = int(counts[batched].max())
        k = min(nc, negs_num_per_query + most)
        index = IndexFlatL2(xq.shape[1], device=dev)
        index.add(xb.index_select(0, cand_t))
        bt = torch.from_numpy(batched).to(dev)
        xqb = xq if batched.size == nq else xq.index_select(0, bt)
        _, I = index.search_device(xqb, k)                                      # [Qb,k] positions in cand, nearest first
        nb = batched.size
        if most:
            pos_q = np.concatenate([np.full(hits_per_q[q].size, i, dtype=np.int64) for i, q in enumerate(batched)])
            pos_c = np.concatenate([hits_per_q[q] for q in batched])
            keys = torch.from_numpy(pos_q * nc + pos_c).to(dev)
            rowkey = torch.arange(nb, device=dev).view(-1, 1) * nc + I
            keep = ~torch.isin(rowkey, keys)
        else:
            keep = torch.ones_like(I, dtype=torch.bool)
        rank = torch.cumsum(keep, 1) - 1                                        # rank among kept entries
        sel = keep & (rank < negs_num_per_query)
        rows = bt.view(-1, 1).expand_as(I)
        out[rows[sel], rank[sel]] = cand_t[I[sel]]
    for q in single:
        mask = np.ones(nc, dtype=bool)
        mask[hits_per_q[q]] = False
        sub = cand[mask]                                                        # sampled order, soft positives removed
        sub_t = torch.from_numpy(sub).to(dev)
        index = IndexFlatL2(xq.shape[1], device=dev)
        index.add(xb.index_select(0, sub_t))
        _, I = index.search_device(xq[q:q + 1], negs_num_per_query)
        out[q] = sub_t[I[0]]
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


def compute_triplets_partial_sharded(query_features, database_features, sampled_queries_indexes, hard_positives_per_query,
                                     soft_positives_per_query, sampled_database_indexes, negs_num_per_query=10, device="cuda"):
    """Data-parallel cache refresh (the loop it shards: reference datasets_ws_nuscenes.py:1398-1410).
    Every rank holds the whole feature cache (the all-gather of the sharded cache extraction is the exchange step) and
    the same sampled index arrays (same seed); rank r mines its contiguous shard of the sampled queries
    (parallel.shard_range) and the [Q_r, 2 + negs] tables are all-gathered: every rank returns the full table, equal to
    compute_triplets_partial on one rank."""
    from . import parallel
    import torch.distributed as dist
    sq = np.asarray(sampled_queries_indexes, dtype=np.int64)
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = parallel.shard_range(len(sq), rank, world)
    qf = query_features[lo:hi]
    local, err = None, None
    try:
        if hi > lo:
            local = compute_triplets_partial(qf, database_features, sq[lo:hi], hard_positives_per_query, soft_positives_per_query,
                                             sampled_database_indexes, negs_num_per_query, device)
        else:
            local = torch.zeros((0, 2 + negs_num_per_query), dtype=torch.int64, device=torch.device(device))
    except ValueError as e:          # e.g. too few candidate negatives for one of THIS shard's queries only
        err = f"rank {rank}: {e}"
    # a failure on one rank must fail every rank BEFORE the gather (the others would wait in the collective for ever)
    errs = [e for e in parallel.all_gather_object(err) if e]
    if errs:
        raise ValueError("compute_triplets_partial_sharded: " + "; ".join(errs))
    return parallel.all_gather_rows(local)
