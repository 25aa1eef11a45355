"""CPU restatement of the reference's training losses (TEST INFRASTRUCTURE ONLY).

    compute_loss         train.py:51-79   (criterion == "triplet": nn.TripletMarginLoss(margin, p=2,
                                           reduction="sum") over the 10 negative index views, / (B*negs))
    compute_other_loss   compute_other_loss.py:21-113

Distances are evaluated directly (sqrt of the summed squared differences) in the input dtype; the
reference's torch.cdist switches to the |x|^2+|y|^2-2xy form above 25 rows, which differs by fp32
rounding only (the golden fixture from the reference module pins both to 1e-4).
"""
import torch
import torch.nn.functional as F


def _cdist(x, y):
    d = x.unsqueeze(1) - y.unsqueeze(0)
    sq = (d * d).sum(-1)
    # sqrt with a zero (sub)gradient at 0, as torch.cdist's backward
    safe = torch.where(sq > 0, sq, torch.ones_like(sq))
    return torch.where(sq > 0, safe.sqrt(), torch.zeros_like(sq))


def compute_bcemat(eastnorthdist_mat, positive_thd=10, negative_thd=25):
    """compute_other_loss.py:21-26"""
    m = torch.zeros_like(eastnorthdist_mat) - 1
    m[eastnorthdist_mat < positive_thd] = 0
    m[eastnorthdist_mat > negative_thd] = 1
    return m


def _term(featsdist, bcemat, otherloss_type):
    """compute_other_loss.py:31-53"""
    mask = bcemat != -1
    z, t = featsdist[mask], bcemat[mask]
    if otherloss_type == "bce":
        return F.binary_cross_entropy_with_logits(z, t)
    if otherloss_type == "mse":
        return F.mse_loss(torch.sigmoid(z), t)
    if otherloss_type == "l1":
        return F.l1_loss(torch.sigmoid(z), t)
    raise NotImplementedError(otherloss_type)


def compute_other_loss(feats_ground, feats_aerial, data_dict, positive_thd=10, negative_thd=25,
                       otherloss_type="bce", otherloss_weight=0.01):
    """compute_other_loss.py:56-113"""
    g_embed, g_img, g_vox = feats_ground["embedding"], feats_ground["imagevec_org"], feats_ground["voxvec_org"]
    a = feats_aerial["embedding"]
    b, ndb, c = a.shape
    a = a.reshape(-1, c)
    en_g = data_dict["query_eastnorth"].to(a.dtype)
    en_a = data_dict["db_eastnorth"].reshape(-1, 2).to(a.dtype)
    en_ag = torch.cat([en_a, en_g], 0)
    loss = _term(_cdist(a, a), compute_bcemat(_cdist(en_a, en_a), positive_thd, negative_thd), otherloss_type)
    for g in (g_embed, g_img, g_vox):
        loss = loss + _term(_cdist(g, torch.cat([a, g], 0)),
                            compute_bcemat(_cdist(en_g, en_ag), positive_thd, negative_thd), otherloss_type)
    return loss * otherloss_weight


def compute_loss(triplets_local_indexes, features, train_batch_size, negs_num_per_query, margin):
    """train.py:51-61,76-77 (criterion == "triplet")"""
    loss = 0
    t = torch.transpose(triplets_local_indexes.view(train_batch_size, negs_num_per_query, 3), 1, 0)
    for triplets in t:
        qi, pi, ni = triplets.T
        loss = loss + F.triplet_margin_loss(features[qi], features[pi], features[ni], margin=margin, p=2,
                                            reduction="sum")
    return loss / (train_batch_size * negs_num_per_query)


def compute_loss_sare(triplets_local_indexes, features, train_batch_size, negs_num_per_query, criterion):
    """train.py:62-77 with model/functional.py:5-27: per group (10 rows for 'sare_joint', 1 for 'sare_ind') the query
    and positive of the group's first row against the group's negatives, -log_softmax(-squared distances)[0]."""
    t = triplets_local_indexes.view(-1, 3)
    group = 10 if criterion == "sare_joint" else 1
    loss = 0
    for bt in t.view(-1, group, 3):
        q, p, n = features[bt[0, 0]].unsqueeze(0), features[bt[0, 1]].unsqueeze(0), features[bt[:, 2]]
        dist = -torch.cat((((q - p) ** 2).sum(1), ((q - n) ** 2).sum(1)))
        loss = loss - F.log_softmax(dist, 0)[0]
    return loss / (train_batch_size * negs_num_per_query)
