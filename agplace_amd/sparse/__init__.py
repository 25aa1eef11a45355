"""Sparse-voxel branch of the query network: MinkowskiEngine-shaped modules on
the HIP gather-GEMM.  See coords.py (sparse tensor + coordinate maps) and modules.py."""
from .coords import SparseTensor  # noqa: F401
from .modules import (ECABasicBlock, ECALayer, MinkFPN, MinkGeM, MinkowskiBatchNorm,  # noqa: F401
                      MinkowskiConvolution, MinkowskiConvolutionTranspose)
