"""CPU oracle for the AGPlace hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a from-scratch CPU restatement (plain PyTorch fp32/fp64 and
numpy) of the arithmetic on the path SURVEY.md section 8 scopes:

    ImageFE (ResNet stem + layer1..3)      reference network_mm/image_fe.py:97-113
    GeM pooling                            reference network_mm/image_pooling.py:8-16
    FC / ODEFunc / FCODE / DiffBlock       reference network_mm/ffns.py:14-87, diff_block.py:18-49
    FuseBlockToShallow                     reference network_mm/fuse_block_toshallow.py:79-121
    Stage2FuseBlockAdd (image side)        reference network_mm/stage2fuse_blockadd.py:180-219
    MM.forward_q glue                      reference network_mm/mm.py:70-160
    DBVanilla2D.forward_db                 reference models_baseline/dbvanilla2d.py:50-101
    NetVLAD.forward                        reference model/aggregation.py:126-146
    IndexFlatL2 search + compute_recall    reference test.py:24-84

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it, and only as the checker.  The product package (agplace_amd) never imports
it and has no CPU fallback.

PARITY PINNING STATUS
---------------------
* Pinned by golden vectors generated from the reference's own Python classes
  (tests/golden/make_golden.py imports /root/reference under sys.modules stubs):
  FC, GeM (3 copies), Basic, BasicBlock, FFNFuse, NetVLAD, DiffBlock/FCODE
  *wiring*, DBVanilla2D.MLP, functional.gem, compute_recall's recall arithmetic.
* PARITY UNPINNED (arithmetic lives in third-party packages that are absent
  from /root/reference and from this image; restated from their published
  algorithms, anchored by analytic known-answer tests in tests/):
    - torchdiffeq (unpinned version, README.md:42): fixed-grid euler / midpoint /
      rk4 (3/8 rule) step functions and grid constructor  -> oracle/ode.py
    - faiss-cpu (unpinned, README.md:48): IndexFlatL2.search -> oracle/knn.py
    - torchvision==0.15.1 ResNet definition (README.md:19)   -> oracle/resnet.py
    - MinkowskiEngine sparse branch: out of scope, the
      oracle takes the pooled vox vectors as inputs.
"""
from . import ode, resnet, nets, knn  # noqa: F401
