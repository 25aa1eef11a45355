"""Hot-path configuration (the subset of the reference's `opt` the path reads).

The reference evaluates one global argparse namespace at import time in every module
(`opt = parse_arguments()`, e.g. network_mm/ffns.py:10-11).  Here the same fields, with the
same names and defaults (reference tools/options.py:28-58,85-155,221), live in an explicit
dataclass; `from_reference_opt(ns)` adapts a reference namespace.  `set_options()` installs
the process-wide default that modules pick up when no explicit `opt` is passed, which keeps
the reference's zero-argument constructors (`MM()`, `GeM()`, `DiffBlock(dim, ode_dim)`).
"""
from dataclasses import dataclass, field, fields
from typing import List, Optional


@dataclass
class Options:
    # data / retrieval
    maptype: str = "satellite"
    features_dim: int = 256
    train_batch_size: int = 16
    infer_batch_size: int = 32
    negs_num_per_query: int = 10
    neg_samples_num: int = 1000
    recall_values: List[int] = field(default_factory=lambda: [1, 5, 10, 20])
    # database model
    dbimage_fe: str = "resnet18"
    dbimage_fe_layers: str = "2_2_2"
    share_dbfe: bool = False
    # query model
    mm_imgfe: str = "resnet18"
    mm_imgfe_layers: str = "2_2_2"
    mm_imgfe_planes: str = "64_128_256"
    mm_imgfe_dim: int = 256
    mm_voxfe_layers: str = "1_1_1"
    mm_voxfe_planes: str = "64_128_256"
    mm_voxfe_ntd: int = 0
    mm_voxfe_dim: int = 256
    mm_bevfe_planes: str = "64_128_256"
    mm_bevfe_dim: int = 256
    mm_stg2fuse_dim: int = 256
    output_type: List[str] = field(default_factory=lambda: ["image", "vox", "shallow"])
    output_l2: bool = True
    final_type: List[str] = field(
        default_factory=lambda: ["imageorg", "voxorg", "shalloworg", "stg2image", "stg2vox"])
    final_fusetype: str = "add"
    final_l2: bool = False
    image_weight: float = 1.0
    image_learnweight: bool = False
    vox_weight: float = 1.0
    vox_learnweight: bool = False
    shallow_weight: float = 1.0
    shallow_learnweight: bool = False
    diff_type: str = "fcode@relu"
    diff_direction: str = "backward"
    odeint_method: str = "euler"
    odeint_size: float = 0.1
    tol: float = 1e-3
    imagevoxorg_weight: float = 0.0
    imagevoxorg_learnweight: bool = False
    shalloworg_weight: float = 1.0
    shalloworg_learnweight: bool = False
    stg2imagevox_weight: float = 0.1
    stg2imagevox_learnweight: bool = False
    stg2fuse_weight: float = 0.0
    stg2fuse_learnweight: bool = False
    stg2nlayers: int = 1
    stg2fuse_type: Optional[str] = "basic"
    stg2_type: str = "full"
    stg2_useproj: bool = True
    # MI355X build: MFMA operand precision of the INFERENCE convolutions (include/agplace_hip.h):
    #   4 = F16 (library default since round 5 = what bench.py measures): fp16 activations x fp16 weights, ONE MFMA product;
    #       descriptors 2.4e-4 .. 3.8e-4, feature maps 4.5e-4 .. 8.5e-4 at the bench size (bar 1e-3;
    #       tests/test_gpu_models.py holds every descriptor under 5e-4, also with checkpoint-like statistics); the 64-channel
    #       BasicBlocks of layer 1 run as one fused kernel (csrc/fblock64.hip)
    #   2 = F16W2 (the opt-in "tight" mode; the default of rounds 1-4): fp16 activations x fp16 hi + (e4m3) lo weights, 1.5 MFMA
    #       products; descriptors 3e-5 .. 1.6e-4, maps <= 6e-4 relative to fp32 -- the weights' rounding, a coherent
    #       perturbation, is what the lo product removes; about half the throughput of mode 4
    #   3 = BF16X3: split-bf16 activations and weights, three products, ~1e-5 everywhere, fp32 range
    # Modes 2 and 4 store fp16 maps, which saturate at +-65504: the first inference forward after a weight (re)load counts
    # saturated map elements and warns (resnet.SATURATION_CHECK) -- such a checkpoint needs mode 3.  Training (.train()) always
    # runs on split-bf16 maps (3); kNN has its own setting.
    mfma_precision: int = 4
    # TRAINING (.train(), or .eval() with gradients): 32 (default) = the tight mode -- split-bf16 maps, three MFMA products in
    # every forward and data-gradient conv, one fp16 product in the weight gradients; every parameter gradient within 1e-3 of
    # fp64 autograd (tests/test_gpu_train.py).  16 = the opt-in FAST mode: the forward of the 3x3 stride-1 convs as ONE
    # fp16 x fp16 product (train_graph.FWD_F16; fp32 master weights, fp64-finalised statistics, three-product data gradients);
    # gradient error and cosine against the fp64 oracle: tests/test_gpu_train.py::test_fast_training_mode_gradients, timing:
    # bench.py train.fast_mode.
    train_precision: int = 32
    # 3 (default) | 1 = opt-in: the data gradients of the 3x3 convs as ONE bf16 product of the hi planes (train_graph.DGRAD_HI_ONLY),
    # usually together with train_precision = 16; measured error: tests/test_gpu_train.py, timing: bench.py train.fast_mode
    train_dgrad_products: int = 3
    # inference: MM.forward embeds a batch as this many sub-batches on as many HIP streams (1 = off)
    query_substreams: int = 1
    # inference: the vector path (everything after the backbones) as two program launches (agplace_amd/vecprog.py) instead
    # of one launch per Linear / FCODE / LayerNorm / normalize / weighted sum; False selects the per-op path
    fused_vector_path: bool = True
    knn_precision: int = 4      # coarse pass: 4 = fp16 (default, fastest), 3 = split-bf16, 1 = bf16; the result is exact in all
    # losses (tools/options.py:158-159,169,189,48,35)
    otherloss_type: str = "bce"
    otherloss_weight: float = 0.01
    tripletloss_weight: float = 1.0
    margin: float = 0.1
    criterion: str = "triplet"
    negs_num_per_query: int = 10
    train_batch_size: int = 16
    train_positives_dist_threshold: int = 10      # tools/options.py:45
    val_positive_dist_threshold: int = 25         # tools/options.py:44

    def copy(self, **kw):
        d = {f.name: getattr(self, f.name) for f in fields(self)}
        d.update(kw)
        return Options(**d)


def from_reference_opt(ns) -> Options:
    """Adapt a reference argparse namespace (tools/options.py) to Options."""
    o = Options()
    for f in fields(Options):
        if hasattr(ns, f.name):
            v = getattr(ns, f.name)
            if f.name in ("output_type", "final_type") and isinstance(v, str):
                v = v.split("_")
            setattr(o, f.name, v)
    return o


_default = Options()


def get_options() -> Options:
    return _default


def set_options(o: Options) -> Options:
    global _default
    _default = o
    return o
