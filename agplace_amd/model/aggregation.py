"""NetVLAD and GeM aggregators, drop-ins for the two classes of reference model/aggregation.py the
hot path names (NetVLAD :96-146, GeM); MAC/SPoC/RMAC/RRM/CRN are unused by the reference's live
path and are not built.

NetVLAD.forward(x[N,D,H,W]) -> [N, K*D]: L2-normalise over D, 1x1 conv soft-assignment, softmax
over clusters, residual aggregation, intra-normalisation, flatten, L2-normalise -- one HIP kernel
(agp_netvlad_fwd).  state_dict keys: conv.weight [K,D,1,1] (bias=False), centroids [K,D].
`init_params` / `initialize_netvlad_layer` (host-side numpy + faiss k-means over sampled descriptors, :112-124,148-174)
are one-off training-time utilities outside the hot path and are not provided: load
`conv.weight` / `centroids` through the state_dict.
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

    def forward(self, x):
        with torch.no_grad():
            return ops.netvlad(x, self.conv.weight, self.centroids, self.normalize_input)
