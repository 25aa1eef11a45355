"""CPU restatement of the reference's per-query triplet mining (TEST INFRASTRUCTURE ONLY).

Follows datasets/datasets_ws_nuscenes.py: get_best_positive_index (:1241-1248),
get_hardest_negatives_indexes (:1250-1258) and the loop body of compute_triplets_partial_sep
(:1394-1404).  faiss.IndexFlatL2 is absent from this image (PARITY UNPINNED, see oracle/knn.py): it is
restated as an exact fp64 brute force with the earlier candidate winning ties.
"""
import numpy as np

from . import knn


def best_positive_index(query_features, cache, hard_positives):
    pos = np.asarray(hard_positives, dtype=np.int64)
    _, I, _ = knn.knn_l2_fp64(query_features.reshape(1, -1), cache[pos], 1)
    return int(pos[I[0, 0]])


def hardest_negatives_indexes(query_features, cache, neg_samples, negs_num_per_query):
    neg_samples = np.asarray(neg_samples, dtype=np.int64)
    _, I, _ = knn.knn_l2_fp64(query_features.reshape(1, -1), cache[neg_samples], negs_num_per_query)
    return neg_samples[I.reshape(-1)].astype(np.int64)


def compute_triplets_partial(query_features, database_features, sampled_queries_indexes, hard_positives_per_query,
                             soft_positives_per_query, sampled_database_indexes, negs_num_per_query=10):
    rows = []
    for i, query_index in enumerate(sampled_queries_indexes):
        qf = query_features[i]
        best = best_positive_index(qf, database_features, hard_positives_per_query[query_index])
        soft = soft_positives_per_query[query_index]
        neg_indexes = np.setdiff1d(sampled_database_indexes, soft, assume_unique=True)
        negs = hardest_negatives_indexes(qf, database_features, neg_indexes, negs_num_per_query)
        rows.append((int(query_index), best, *[int(v) for v in negs]))
    return np.asarray(rows, dtype=np.int64)
