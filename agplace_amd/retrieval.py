"""Exact L2 retrieval: faiss-shaped IndexFlatL2 and the reference's compute_recall.

Replaces `faiss.IndexFlatL2(d)`, `.add(xb)`, `.search(xq, k) -> (D, I)` at reference
test.py:27-32 and datasets/datasets_ws_nuscenes.py:1241-1258, and `compute_recall`
(test.py:24-84, test_method='hard_resize').  Distances are SQUARED L2 (float32), ascending,
labels int64, (FLT_MAX, -1) beyond ntotal -- faiss's conventions.  The search is the gfx950
MFMA kernel pipeline of agp_knn_search; results are exact (fp64 re-evaluation of a provably
complete candidate set), ties ordered by ascending database index.
"""
import numpy as np
import torch

from . import _lib
from ._lib import ptr, check
from .options import get_options


class IndexFlatL2:
    def __init__(self, d, device="cuda", prec=None):
        """Any width d, as faiss takes (test.py:27): the kernels work on multiples of 32 columns, so other widths are zero-padded
        on the device (zero columns change no distance)."""
        if d < 1:
            raise ValueError("IndexFlatL2: d must be positive")
        self.d = d
        self.dpad = (d + 31) // 32 * 32
        self.device = torch.device(device)
        self.prec = prec or get_options().knn_precision
        self.ntotal = 0
        self._buf = None            # [capacity, dpad] fp32; rows [0, ntotal) are the database (capacity doubles: adds are O(n) in all)
        self._prepared = None
        self._ws = {}               # search workspace per launching stream (grow-only)

    @property
    def _xb(self):
        return None if self._buf is None else self._buf[:self.ntotal]

    # ---- faiss API
    def add(self, xb):
        """faiss's incremental add (test.py fills an index batch by batch if used that way): rows are appended into a buffer whose
        capacity doubles, so n rows added in any number of calls cost O(n) copies (a torch.cat per call was O(n^2))."""
        xb = self._to_dev(xb)
        n = xb.shape[0]
        if n == 0:
            return
        if self._buf is None and self.ntotal == 0:
            self._buf = xb                  # the common single add: adopt the tensor, no copy
        else:
            need = self.ntotal + n
            if need > self._buf.shape[0]:
                grown = torch.empty((max(need, 2 * self._buf.shape[0]), self.dpad), dtype=torch.float32, device=self.device)
                grown[:self.ntotal] = self._buf[:self.ntotal]
                self._buf = grown
            self._buf[self.ntotal:need] = xb
        self.ntotal += n
        self._prepared = None

    def reset(self):
        self._buf, self.ntotal, self._prepared = None, 0, None

    def _workspace(self, nbytes):
        key = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    def search(self, xq, k):
        """(D float32 [nq,k], I int64 [nq,k]); numpy in -> numpy out, torch in -> torch out."""
        as_numpy = isinstance(xq, np.ndarray)
        D, I = self.search_device(self._to_dev(xq), k)
        if as_numpy:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    # ---- device path
    def _to_dev(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(self.device, dtype=torch.float32)
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"IndexFlatL2: expected [n, {self.d}] vectors, got {tuple(x.shape)}")
        if self.dpad != self.d:
            x = torch.nn.functional.pad(x, (0, self.dpad - self.d))
        return x.contiguous()

    def _prepare(self):
        if self._prepared is None:
            L = _lib.load()
            nb = self.ntotal
            nb_pad = L.agp_knn_pad_rows(nb)
            hi = torch.empty((nb_pad, self.dpad), dtype=torch.bfloat16, device=self.device)
            lo = torch.empty((nb_pad, self.dpad), dtype=torch.bfloat16, device=self.device) if self.prec == 3 else None
            norm = torch.empty(nb_pad + 32, dtype=torch.float32, device=self.device)
            check(L.agp_knn_prepare_db(ptr(self._xb), nb, self.dpad, self.prec, ptr(hi), ptr(lo), ptr(norm),
                                       _lib.stream()), "agp_knn_prepare_db")
            self._prepared = (hi, lo, norm)
        return self._prepared

    def search_device(self, xq, k):
        """xq: a device tensor [nq, d] (or already padded [nq, dpad])."""
        L = _lib.load()
        if xq.shape[1] != self.dpad:
            xq = self._to_dev(xq)
        nq = xq.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        if nq == 0:
            return D, I
        if self.ntotal == 0:
            D.fill_(3.4028234663852886e38)
            I.fill_(-1)
            return D, I
        hi, lo, norm = self._prepare()
        # bound the [groups x queries] workspace: chunk the queries
        chunk = max(1, min(nq, int(2 ** 31 // max(self.ntotal // 4, 1))))
        for s in range(0, nq, chunk):
            q = xq[s:s + chunk]
            nbytes = L.agp_knn_workspace_bytes(q.shape[0], self.ntotal, self.dpad, k)
            ws = self._workspace(nbytes)
            check(L.agp_knn_search(ptr(q), q.shape[0], ptr(self._xb), ptr(hi), ptr(lo), ptr(norm),
                                   self.ntotal, self.dpad, k, self.prec, ptr(D[s:s + chunk]),
                                   ptr(I[s:s + chunk]), ptr(ws), nbytes, _lib.stream()), "agp_knn_search")
        return D, I


    def coarse_pass_device(self, xq):
        """The search's first stage alone (agp_knn_coarse_pass: query preparation + the coarse distance pass into a workspace),
        no result: lets a caller time the search's dominant kernel with events on the launch stream (bench.py's kNN roofline)."""
        L = _lib.load()
        if xq.shape[1] != self.dpad:
            xq = self._to_dev(xq)
        hi, lo, norm = self._prepare()
        nbytes = L.agp_knn_workspace_bytes(xq.shape[0], self.ntotal, self.dpad, 1)
        ws = self._workspace(nbytes)
        check(L.agp_knn_coarse_pass(ptr(xq), xq.shape[0], ptr(hi), ptr(lo), ptr(norm), self.ntotal, self.dpad, self.prec, ptr(ws),
                                    nbytes, _lib.stream()), "agp_knn_coarse_pass")


def recall_from_predictions(args, predictions, test_ds):
    """The recall arithmetic of reference test.py:73-83: predictions int64 [Q, max(recall_values)]."""
    if not isinstance(predictions, np.ndarray):
        predictions = predictions.cpu().numpy()
    positives_per_query = test_ds.get_positives()
    recalls = np.zeros(len(args.recall_values))
    for query_index, pred in enumerate(predictions):
        for i, n in enumerate(args.recall_values):
            if np.any(np.isin(pred[:n], positives_per_query[query_index])):
                recalls[i:] += 1
                break
    recalls = recalls / test_ds.queries_num * 100
    recalls_str = ", ".join([f"R@{val}: {rec:.1f}" for val, rec in zip(args.recall_values, recalls)])
    return recalls, recalls_str


def merge_crops(args, distances, predictions, test_ds, test_method):
    """The five-crop post-processing of reference test.py:35-70 (DVGLB's 'nearest_crop' / 'maj_voting' test methods: every query
    contributes 5 consecutive descriptor rows; the 5 x k candidate lists of a query are merged into its k nearest DISTINCT
    database rows, 'maj_voting' after lowering the distance of candidates that several crops agree on by
    majority_weight * count / n inside the top-1 / top-5 / top-10 columns).  Host arithmetic on [5 Q, k] arrays."""
    k, nq, ncrop = max(args.recall_values), test_ds.queries_num, 5
    d = np.array(distances, dtype=np.float32).reshape(nq, ncrop, k)
    p = np.array(predictions).reshape(nq, ncrop, k)
    out = np.empty((nq, k), dtype=p.dtype)
    for q in range(nq):
        dq, pq = d[q], p[q]
        if test_method == 'maj_voting':
            for n in (1, 5, 10):
                head_p, head_d = pq[:, :n], dq[:, :n]              # views: the next round sees this round's votes
                vals, counts = np.unique(head_p, return_counts=True)
                for val, cnt in zip(vals[counts > 1], counts[counts > 1]):
                    head_d[head_p == val] -= args.majority_weight * cnt / n
        order = np.argsort(dq.reshape(-1))                          # (the reference's default sort, on the same values)
        cand = pq.reshape(-1)[order]
        _, first = np.unique(cand, return_index=True)               # first = closest occurrence of every distinct row
        out[q] = cand[np.sort(first)][:k]
    return out


def compute_recall(args, queries_features, database_features, test_ds, test_method='hard_resize'):
    """reference test.py:24-84.  `args` needs features_dim and recall_values (+ majority_weight for 'maj_voting'); `test_ds` needs
    get_positives() and queries_num, exactly as in the reference.  'nearest_crop' / 'maj_voting' expect 5 descriptor rows per
    query (DVGLB's five-crop test methods, test.py:35-70)."""
    if test_method not in ('hard_resize', 'nearest_crop', 'maj_voting'):
        raise NotImplementedError(test_method)
    index = IndexFlatL2(args.features_dim)
    index.add(database_features)
    distances, predictions = index.search(queries_features, max(args.recall_values))
    if test_method != 'hard_resize':
        if not isinstance(predictions, np.ndarray):
            distances, predictions = distances.cpu().numpy(), predictions.cpu().numpy()
        predictions = merge_crops(args, distances, predictions, test_ds, test_method)
    return recall_from_predictions(args, predictions, test_ds)


def distributed_search(local_queries, local_database, k, device="cuda", prec=None, timings=None):
    """Data-parallel exact kNN (the loops it shards: reference test.py:125-176).
    Rank r holds the descriptors of ITS contiguous shard of the database rows and of the query rows
    (parallel.shard_range over the dataset order, which is how a sharded extraction loop fills them).  The database
    shards are all-gathered -- the one exchange step, [N,256] fp32 over xGMI -- so every rank holds the full database;
    each rank then searches only its own queries, and the [Q_r, k] results are all-gathered (Q k 12 bytes).
    Returns (D [Q,k] float32, I [Q,k] int64) for ALL queries in dataset order, on every rank.
    timings (optional dict): receives the wall time of the phases in ms, each bracketed by a device synchronisation --
    allgather_ms (+ allgather_bytes = the bytes this rank RECEIVED), prepare_ms (index planes), search_ms,
    gather_results_ms -- and the built `index` (bench.py's N > 1 kNN leg keeps searching it)."""
    import time
    from . import parallel
    dev = torch.device(device)

    def to_dev(x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return x.to(dev, dtype=torch.float32).contiguous()

    def mark():
        if timings is None:
            return 0.0
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return time.perf_counter()
    ldb, lq = to_dev(local_database), to_dev(local_queries)
    t0 = mark()
    db = parallel.all_gather_rows(ldb)
    t1 = mark()
    index = IndexFlatL2(db.shape[1], device=dev, prec=prec)
    index.add(db)
    if timings is not None and hasattr(index, "_prepare"):
        index._prepare()
    t2 = mark()
    D, I = index.search_device(lq, k)
    t3 = mark()
    D, I = parallel.all_gather_rows(D), parallel.all_gather_rows(I)
    t4 = mark()
    if timings is not None:
        timings.update(allgather_ms=(t1 - t0) * 1e3, allgather_bytes=int((db.shape[0] - ldb.shape[0]) * db.shape[1] * 4),
                       prepare_ms=(t2 - t1) * 1e3, search_ms=(t3 - t2) * 1e3, gather_results_ms=(t4 - t3) * 1e3, index=index)
    return D, I


def distributed_compute_recall(args, local_queries_features, local_database_features, test_ds, test_method='hard_resize',
                               device="cuda"):
    """compute_recall (reference test.py:24-84) over descriptors that were extracted data-parallel: every rank passes the
    rows of its own shard (see distributed_search); every rank returns the same (recalls, recalls_str)."""
    if test_method not in ('hard_resize', 'nearest_crop', 'maj_voting'):
        raise NotImplementedError(test_method)
    distances, predictions = distributed_search(local_queries_features, local_database_features, max(args.recall_values), device=device)
    if test_method != 'hard_resize':
        predictions = merge_crops(args, distances.cpu().numpy(), predictions.cpu().numpy(), test_ds, test_method)
    return recall_from_predictions(args, predictions, test_ds)
