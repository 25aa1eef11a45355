"""NetVLAD and GeM aggregators, drop-ins for the two classes of reference model/aggregation.py the
hot path names (NetVLAD :96-146, GeM); MAC/SPoC/RMAC/RRM/CRN are unused by the reference's live
path and are not built.

NetVLAD.forward(x[N,D,H,W]) -> [N, K*D]: L2-normalise over D, 1x1 conv soft-assignment, softmax
over clusters, residual aggregation, intra-normalisation, flatten, L2-normalise -- one HIP kernel
(agp_netvlad_fwd); differentiable in x, conv.weight and centroids (agp_netvlad_bwd).  state_dict keys: conv.weight [K,D,1,1] (bias=False), centroids [K,D].
`init_params(centroids, descriptors)` (:112-124) is provided (code ported from the reference calls it): the
soft-assignment weights from k-means centroids; `initialize_netvlad_layer` (:148-174: dataset sampling + faiss k-means) is a
one-off training-time utility outside the hot path and is not.
"""
import torch
import torch.nn as nn

from .. import ops
from ..network_mm.image_pooling import gem_op


class GeM(nn.Module):
    def __init__(self, p=3, eps=1e-6, work_with_tokens=False):
        super().__init__()
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps
        if work_with_tokens:
            raise NotImplementedError

    def forward(self, x):
        return gem_op(x, self.p, self.eps).view(x.size(0), x.size(1), 1, 1)


class NetVLAD(nn.Module):
    def __init__(self, clusters_num=64, dim=128, normalize_input=True, work_with_tokens=False):
        super().__init__()
        self.clusters_num = clusters_num
        self.dim = dim
        self.alpha = 0
        self.normalize_input = normalize_input
        self.work_with_tokens = work_with_tokens
        if work_with_tokens:
            raise NotImplementedError
        self.conv = nn.Conv2d(dim, clusters_num, kernel_size=(1, 1), bias=False)
        self.centroids = nn.Parameter(torch.rand(clusters_num, dim))

    def init_params(self, centroids, descriptors):
        """Soft-assignment initialisation from cluster centres (reference model/aggregation.py:112-124): with the centres normalised
        to unit length, alpha = -log(0.01) / mean over descriptors of (best - second-best cosine score); conv.weight = alpha * the
        unit centres, centroids = the centres.  `centroids` [K, D] and `descriptors` [n, D]: arrays or tensors (host arithmetic)."""
        c = torch.as_tensor(centroids, dtype=torch.float32).detach().cpu()
        d = torch.as_tensor(descriptors, dtype=torch.float32).detach().cpu()
        if c.dim() != 2 or c.shape != (self.clusters_num, self.dim) or d.dim() != 2 or d.shape[1] != self.dim or self.clusters_num < 2:
            raise ValueError("NetVLAD.init_params: centroids [clusters_num, dim] (clusters_num >= 2) and descriptors [n, dim]")
        unit = c / c.norm(dim=1, keepdim=True)
        top2 = torch.topk(unit @ d.t(), 2, dim=0).values             # [2, n]: best and second-best score of every descriptor
        self.alpha = float(-torch.log(torch.tensor(0.01, dtype=torch.float64)) / (top2[0] - top2[1]).double().mean())
        dev = self.centroids.device
        self.centroids = nn.Parameter(c.clone().to(dev))
        self.conv.weight = nn.Parameter((self.alpha * unit).view(self.clusters_num, self.dim, 1, 1).to(dev))
        self.conv.bias = None

    def forward(self, x):
        return ops.netvlad(x, self.conv.weight, self.centroids, self.normalize_input)
