"""ctypes binding of libagplace_hip.so (the C ABI declared in include/agplace_hip.h).

The product path has NO CPU fallback: if the shared library is missing or an entry
point is absent, importing a kernel-backed op raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGP_HIP_LIB: an alternative build of the same library (tools/census.py uses one compiled with -DAGP_CENSUS=1)
LIB_PATH = os.environ.get("AGP_HIP_LIB") or os.path.join(_HERE, "lib", "libagplace_hip.so")

AGP_OK = 0
PREC_BF16 = 1
PREC_F16W2 = 2
PREC_BF16X3 = 3
PREC_F16 = 4
FMT_BF16, FMT_F16 = 0, 1
ACT = {None: 0, "id": 0, "relu": 1, "tanh": 2, "sigmoid": 3}
ODE = {"euler": 0, "midpoint": 1, "rk4": 2}
E_UNSUPPORTED = 3
GP_FLOATS = 1 + 16384 + 1     # AGP_GP_FLOATS (include/agplace_hip.h): the dL/dp buffer of the GeM backward entries
_ERR = {1: "AGP_E_BADARG (unsupported shape / enum / null pointer)",
        2: "AGP_E_LAUNCH (HIP launch failed)",
        3: "AGP_E_UNSUPPORTED"}


class ConvDesc(C.Structure):
    """struct agp_conv_desc (include/agplace_hip.h)."""
    _fields_ = [
        ("in_hi", C.c_void_p), ("in_lo", C.c_void_p),
        ("w_hi", C.c_void_p), ("w_lo", C.c_void_p),
        ("out_hi", C.c_void_p), ("out_lo", C.c_void_p),
        ("res_hi", C.c_void_p), ("res_lo", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("n", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32), ("cin", C.c_int32),
        ("pin", C.c_int32), ("in_w_step", C.c_int32),
        ("hout", C.c_int32), ("wout", C.c_int32), ("cout", C.c_int32), ("pout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("relu", C.c_int32), ("prec", C.c_int32),
        ("w_q8", C.c_void_p), ("w_q8_exp", C.c_int32),
        ("stat_partial", C.c_void_p),
        ("pool_partial", C.c_void_p), ("pool_p", C.c_void_p), ("pool_eps", C.c_float), ("hi_only", C.c_int32),
        ("w_cm", C.c_void_p),
        ("bstat_z_hi", C.c_void_p), ("bstat_z_lo", C.c_void_p), ("bstat_y_hi", C.c_void_p),
        ("bstat_mean", C.c_void_p), ("bstat_rstd", C.c_void_p),
        ("w_cm_lo", C.c_void_p),
        ("in_h16", C.c_void_p), ("out_absmax", C.c_void_p),
        ("pool_stat", C.c_int32), ("reserved0", C.c_int32),
    ]


class BBlock64Desc(C.Structure):
    """struct agp_bblock64_desc (include/agplace_hip.h)."""
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("scale1", C.c_void_p), ("shift1", C.c_void_p), ("scale2", C.c_void_p), ("shift2", C.c_void_p),
                ("pool_partial", C.c_void_p), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("form", C.c_int32)]


class VecProgOp(C.Structure):
    """struct agp_vecprog_op (include/agplace_hip.h)."""
    _fields_ = [("op", C.c_int32), ("dst", C.c_int32), ("r", C.c_int32 * 6), ("k", C.c_int32), ("act", C.c_int32),
                ("aux", C.c_int32), ("n", C.c_int32), ("f0", C.c_float), ("pad", C.c_int32), ("p", C.c_void_p * 6)]


VP_LOAD, VP_STORE, VP_LINEAR, VP_FCODE, VP_L2NORM, VP_LAYERNORM, VP_WSUM = 1, 2, 3, 4, 5, 6, 7
VECPROG_MAXOPS, VECPROG_NREG = 36, 6

_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every symbol include/agplace_hip.h declares
SIGNATURES = {
    "agp_version": (C.c_char_p, []),
    "agp_arch": (C.c_char_p, []),
    "agp_split_f32": (_I, [_P, _P, _P, _L, _I, _P]),
    "agp_split_conv_weight": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agp_split_conv_weight_both": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "agp_pack_f32_to_nhwc": (_I, [_P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agp_pack_f32_to_nhwc4_h16": (_I, [_P, _L, _L, _L, _L, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "agp_pack_u8_cams_to_nhwc": (_I, [_P, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _I, _P, _P, _P]),
    "agp_unpack_nhwc_to_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "agp_map_zero_halo": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "agp_conv2d_fwd": (_I, [C.POINTER(ConvDesc), _P]),
    "agp_conv2d_fwd_grouped": (_I, [C.POINTER(ConvDesc), _I, _P]),
    "agp_bblock64_fwd_grouped": (_I, [C.POINTER(BBlock64Desc), _I, _P]),
    "agp_bblock64_pool_floats": (_L, [C.POINTER(BBlock64Desc)]),
    "agp_bblock64_pool_finish": (_I, [_P, _I, _I, _I, _P, _P]),
    "agp_conv_w_q8_prepare": (_I, [_P, _I, _I, _P, C.POINTER(C.c_int32), _P]),
    "agp_conv2d_stat_tiles": (_I, [C.POINTER(ConvDesc)]),
    "agp_conv2d_pool_blocks": (_I, [C.POINTER(ConvDesc)]),
    "agp_pool_from_conv": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "agp_bn_stats_from_partial": (_I, [_P, _I, _I, _L, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "agp_stem_pool_fwd": (_I, [C.POINTER(ConvDesc), _P]),
    "agp_stem_pool_raw_fwd": (_I, [C.POINTER(ConvDesc), _I, _L, _L, _L, _L, _I, C.POINTER(_F), C.POINTER(_F), _P]),
    "agp_maxpool3x3s2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _P]),
    "agp_bcast_add_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "agp_pool_workspace_floats": (_L, [_I, _I, _I, _I]),
    "agp_pool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P]),
    "agp_pool_f32_fwd": (_I, [_P, _L, _L, _L, _L, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P]),
    "agp_gem_f32_bwd": (_I, [_P, _L, _L, _L, _L, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P, _P]),
    "agp_linear_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "agp_fcode_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(_F), _I, _P, _P, _P]),
    "agp_fcode_traj_floats": (_L, [_I, _I, _I]),
    "agp_fcode_bwd_workspace_bytes": (_L, [_I, _I, _I]),
    "agp_fcode_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, C.POINTER(_F), _I, _P, _P, _P, _P, _L, _P]),
    "agp_linear_bwd_workspace_bytes": (_L, [_I, _I, _I]),
    "agp_linear_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "agp_layernorm_bwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P]),
    "agp_l2normalize_bwd": (_I, [_P, _P, _I, _I, _P, _P]),
    "agp_layernorm_fwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _I, _P, _P]),
    "agp_l2normalize_fwd": (_I, [_P, _I, _I, _P, _P]),
    "agp_wsum_fwd": (_I, [_P] * 12 + [_L, _P, _P]),
    "agp_dot_f32": (_I, [_P, _P, _L, _P, _P]),
    "agp_vecprog_run": (_I, [C.POINTER(VecProgOp), _I, _I, _I, C.POINTER(_F), _I, _P]),
    "agp_vecprog_run2": (_I, [C.POINTER(VecProgOp), _I, _I, C.POINTER(VecProgOp), _I, _I, _I, C.POINTER(_F), _I, _P]),
    "agp_conv2d_wgrad_workspace_bytes": (_L, [C.POINTER(ConvDesc)]),
    "agp_conv2d_wgrad": (_I, [C.POINTER(ConvDesc), _P, _P, _L, _P]),
    "agp_conv2d_wgrad_param": (_I, [C.POINTER(ConvDesc), _P, _I, _P, _L, _P]),
    "agp_upsample2_zero": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
    "agp_train_reduce_workspace_floats": (_L, [_I, _I, _I, _I]),
    "agp_bn_stats": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F] + [_P] * 10),
    "agp_map_affine": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "agp_bn_bwd": (_I, [_P] * 9 + [_I] * 6 + [_P] * 9),
    "agp_bn_bwd_from_partial": (_I, [_P, _I] + [_P] * 9 + [_I] * 7 + [_P] * 8),
    "agp_bn_bwd_frozen": (_I, [_P] * 9 + [_I] * 6 + [_P] * 9),
    "agp_bn_sums": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agp_bn_sums_from_partial": (_I, [_P, _I, _I, _L, _P, _P]),
    "agp_bn_stats_from_sums": (_I, [_P, _I, _F, _F] + [_P] * 9),
    "agp_bn_bwd_sums": (_I, [_P] * 8 + [_I] * 6 + [_P] * 5),
    "agp_bn_bwd_apply": (_I, [_P] * 11 + [_I] * 6 + [_P] * 6),
    "agp_bn_frozen_coeffs": (_I, [_P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P]),
    "agp_map_chan_sum": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "agp_map_add": (_I, [_P] * 6 + [_I] * 5 + [_P, _P, _P]),
    "agp_maxpool3x3s2_bwd": (_I, [_P, _P, _P] + [_I] * 8 + [_P, _P, _P]),
    "agp_maxpool_bn_bwd": (_I, [_P, _P, _P, _I, _I, _I] + [_P] * 12 + [_I] * 7 + [_P] * 7),
    "agp_affine_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _P] + [_I] * 5 + [_P, _P] + [_I] * 3 + [_P, _P, _P]),
    "agp_pool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "agp_netvlad_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "agp_netvlad_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "agp_netvlad_workspace_bytes": (_L, [_I, _I, _I, _I]),
    "agp_netvlad_fwd_mfma": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _L, _P]),
    "agp_sparse_conv_fwd": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P]),
    "agp_sparse_tile_taps": (_I, [_P, _L, _I, _L, _P, _P, _P, _L, _P]),
    "agp_sparse_zplane_perm": (_I, [_P, _P, _I, _L, _P, _P]),
    "agp_sparse_kernel_map": (_I, [_P, _L, _P, _L, _P, _I, _P, _P, _P]),
    "agp_sparse_kernel_map_grid": (_I, [_P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P]),
    "agp_sparse_conv_cin1_fwd": (_I, [_P, _L, _P, _L, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P]),
    "agp_sparse_conv0_fwd": (_I, [_P, _L, _P, _P, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _I, _P]),
    "agp_sparse_coords_workspace_bytes": (_L, [_L, _I, _I]),
    "agp_sparse_build": (_I, [_P, _I, _L, _P, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P]),
    "agp_sparse_coarsen": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _L, _P]),
    "agp_seg_pool_fwd": (_I, [_P, _P, _P, _I, _I, _P, _F, _P, _P, _P]),
    "agp_eca_scale_fwd": (_I, [_P, _I, _I, _P, _I, _P, _P]),
    "agp_seg_affine_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "agp_sparse_conv_wgrad_workspace_bytes": (_L, [_L, _I, _I, _I]),
    "agp_sparse_conv_wgrad": (_I, [_P, _P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _L, _P]),
    "agp_sparse_conv_cin1_wgrad": (_I, [_P, _L, _P, _L, _I, _P, _P, _I, _P, _P, _I, _P]),
    "agp_seg_dot_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "agp_eca_scale_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P]),
    "agp_seg_pool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _L, _I, _P, _P, _P, _P]),
    "agp_triplet_loss_workspace_floats": (_L, [_I]),
    "agp_triplet_loss": (_I, [_P, _I, _I, _P, _I, _F, _P, _P, _P, _P]),
    "agp_sare_loss": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P]),
    "agp_pairdist_loss_workspace_floats": (_L, [_I, _I]),
    "agp_pairdist_loss": (_I, [_P, _P, _I, _I, _I, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P]),
    "agp_mine_best_positive": (_I, [_P, _L, _P, _L, _I, _P, _P, _P, _P, _P]),
    "agp_knn_pad_rows": (_L, [_L]),
    "agp_knn_prepare_db": (_I, [_P, _L, _I, _I, _P, _P, _P, _P]),
    "agp_knn_workspace_bytes": (_L, [_L, _L, _I, _I]),
    "agp_knn_search": (_I, [_P, _L, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _L, _P]),
    "agp_knn_coarse_pass": (_I, [_P, _L, _P, _P, _P, _L, _I, _I, _P, _L, _P]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol. Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"agplace_amd: HIP extension not built: {LIB_PATH} is missing. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C agplace_amd/csrc`). "
            "There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so; it must be the process's HIP runtime BEFORE this
    # library is mapped, otherwise the kernels register with /opt/rocm's copy and every launch on a
    # torch stream fails.  Importing torch first makes our DT_NEEDED resolve to the loaded copy.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != AGP_OK:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
