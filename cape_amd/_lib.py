"""ctypes binding of libcape_hip.so (the C-ABI declared in include/cape_hip.h).

The product path has NO fallback: if the shared library is missing this module raises at
import, and every compute entry point raises if no HIP device is present.
"""
import ctypes as C
import os

# torch MUST be imported before the library is loaded: libcape_hip.so needs libamdhip64.so.7 and has
# to bind to the SAME HIP runtime instance PyTorch-ROCm uses (its bundled copy), otherwise streams,
# device pointers and the primary context would belong to two different runtimes (hipErrorNoDevice).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcape_hip.so")

MAX_SRC = 8
ACT = {"none": 0, "leaky": 1, "relu": 2, "tanh": 3}
BIAS_NONE, BIAS_CHANNEL, BIAS_VERTEX = 0, 1, 2


class CapeSrc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_sample_stride", C.c_int64), ("ldx", C.c_int32), ("C", C.c_int32),
        ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("vals", C.c_void_p),
        ("w", C.c_void_p), ("w_rs", C.c_int64), ("w_cs", C.c_int64),
        ("w2", C.c_void_p), ("w2_rs", C.c_int64), ("w2_cs", C.c_int64),
    ]


class CapeCondLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_aff", C.c_void_p), ("coef", C.c_void_p), ("dcoef", C.c_void_p),
                ("gw", C.c_void_p), ("gw_aff", C.c_void_p), ("K", C.c_int32), ("F", C.c_int32)]


class CapeSpmmTerm(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_sample_stride", C.c_int64), ("ldx", C.c_int32),
                ("rowptr", C.c_void_p), ("colidx", C.c_void_p), ("vals", C.c_void_p),
                ("y", C.c_void_p), ("y_sample_stride", C.c_int64), ("ldy", C.c_int32), ("scale", C.c_float),
                ("ell_width", C.c_int32), ("rowmax_out", C.c_void_p)]


class CapeGnParamItem(C.Structure):
    _fields_ = [("dgamma_partial", C.c_void_p), ("dbeta_partial", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("N", C.c_int32), ("C", C.c_int32)]


class CapeBwdPrepItem(C.Structure):
    _fields_ = [("workspace", C.c_void_p), ("N", C.c_int32), ("Mo", C.c_int32), ("F", C.c_int32), ("R", C.c_int32),
                ("dbias", C.c_void_p), ("dcoef", C.c_void_p), ("dcoef_g", C.c_void_p), ("dcoef_sample_stride", C.c_int64),
                ("chunks", C.c_int32)]


class CapeDwItem(C.Structure):
    _fields_ = [("srcs", C.c_void_p), ("nsrc", C.c_int32), ("dz", C.c_void_p), ("dz_sample_stride", C.c_int64),
                ("lddz", C.c_int32), ("dz2", C.c_void_p), ("dz2_mask", C.c_uint32), ("N", C.c_int32), ("Mo", C.c_int32),
                ("F", C.c_int32), ("accumulate", C.c_int32), ("bf16", C.c_int32), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64), ("h2", C.c_void_p)]


class CapeH2Src(C.Structure):
    _fields_ = [("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("w_pitch", C.c_int64),
                ("w2_hi", C.c_void_p), ("w2_lo", C.c_void_p), ("w2_pitch", C.c_int64),
                ("rowmax", C.c_void_p), ("rowmax_w", C.c_int32)]


class CapeH2(C.Structure):
    _fields_ = [("src", C.c_void_p), ("wscale_inv", C.c_void_p), ("w2scale_inv", C.c_void_p),
                ("rowmax_out", C.c_void_p), ("rowmax_out_w", C.c_int32)]


class CapeH2Dw(C.Structure):
    _fields_ = [("src_rowmax", C.c_void_p * MAX_SRC), ("src_rowmax_w", C.c_int32 * MAX_SRC),
                ("dz_rowmax", C.c_void_p), ("dz_rowmax_w", C.c_int32), ("dz2_rowmax", C.c_void_p), ("dz2_rowmax_w", C.c_int32)]


class CapeWpieceItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("Ch", C.c_int32), ("K", C.c_int32), ("F", C.c_int32), ("pair_K", C.c_int32),
                ("pair_w", C.c_void_p),
                ("f_hi", C.c_void_p), ("f_lo", C.c_void_p), ("b_hi", C.c_void_p), ("b_lo", C.c_void_p),
                ("fscale_inv", C.c_void_p), ("bscale_inv", C.c_void_p), ("bscale_c_inv", C.c_void_p),
                ("fpair_w", C.c_void_p), ("fpair_rows", C.c_int32), ("reserved", C.c_int32), ("colmax_partial", C.c_void_p)]


class CapeRank(C.Structure):
    _fields_ = [("R", C.c_int32), ("rowscale", C.c_void_p), ("coef", C.c_void_p), ("to_acc2", C.c_uint32)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "cape_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C cape_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_i32, _i64, _f32, _p = C.c_int32, C.c_int64, C.c_float, C.c_void_p
_SRCP = C.POINTER(CapeSrc)

SIGNATURES = {
    "cape_abi_version": (C.c_int, []),
    "cape_spin_us": (C.c_int, [_i32, _p]),
    "cape_csr_validate": (C.c_int, [_i32, _i32, _i64, _p, _p]),
    "cape_gconv_fwd": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _i32, _i32, _i32, _p, _i32, _i32, _p,
                                 C.POINTER(CapeRank), _i32, _p]),
    "cape_gconv_fwd_plan": (C.c_int, [_SRCP, _i32, _i32, _i32, _i32, C.POINTER(_i32)]),
    "cape_gconv_fwd_h2": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _i32, _i32, _i32, _p, _i32, _i32, _p,
                                    C.POINTER(CapeRank), _i32, C.POINTER(CapeH2), _p]),
    "cape_gconv_fwd_plan_h2": (C.c_int, [_SRCP, _i32, _i32, _i32, _i32, C.POINTER(CapeH2), C.POINTER(_i32)]),
    "cape_gconv_dw_stage_h2": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, _i32, _p, _i64, _i32,
                                         C.POINTER(CapeH2Dw), _p]),
    "cape_gconv_dw_plan_h2": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, C.POINTER(CapeH2Dw),
                                        C.POINTER(_i32)]),
    "cape_rowmax": (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _p, _i32, _p]),
    "cape_weight_pieces_blocks": (C.c_int, [C.c_void_p, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "cape_weight_pieces": (C.c_int, [_p, _i32, _p, _i32, _p, _i32, _p]),
    "cape_rowscale_reduce_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "cape_rowscale_reduce": (C.c_int, [_p, _i64, _i32, _p, _i32, _i32, _i32, _i32, _p, _p, _i64, _p]),
    "cape_gconv_dw_workspace_bytes": (_i64, [_SRCP, _i32, _i32, _i32, _i32]),
    "cape_gconv_dw": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "cape_gconv_dw_stage": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, _i32, _p, _i64, _i32, _p]),
    "cape_condnet_fwd": (C.c_int, [_p, _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cape_condnet_bwd": (C.c_int, [_p, _i32, _p, _i32, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p]),
    "cape_gconv_dw_reduce_batch": (C.c_int, [C.c_void_p, _i32, _p]),
    "cape_gconv_dw_plan": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, C.POINTER(_i32)]),
    "cape_gconv_fwd_bf16": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _i32, _i32, _i32, _p, _i32, _i32, _p,
                                      C.POINTER(CapeRank), _i32, _p]),
    "cape_gconv_fwd_plan_bf16": (C.c_int, [_SRCP, _i32, _i32, _i32, _i32, C.POINTER(_i32)]),
    "cape_gconv_dw_bf16": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "cape_gconv_dw_stage_bf16": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, _i32, _p, _i64, _i32, _p]),
    "cape_gconv_dw_plan_bf16": (C.c_int, [_SRCP, _i32, _p, _i64, _i32, _p, C.c_uint32, _i32, _i32, _i32, C.POINTER(_i32)]),
    "cape_bwd_prep_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "cape_bwd_prep": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _i32, _p, _p, _i64, _i32, _p, _p, _i32, _p, _i32, _p,
                                _i64, _i32, _i32, _i32, _i32, _p, _i64, _p, _p]),
    "cape_bwd_prep_finalize": (C.c_int, [C.c_void_p, _i32, _p]),
    "cape_spmm": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _i32, _i32, _f32, _p, _i64, _i32, _f32, _p, _i64, _i32,
                            _i32, _i32, _i32, _p, _p]),
    "cape_spmm_multi": (C.c_int, [C.POINTER(CapeSpmmTerm), _i32, _i32, _p, _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "cape_spmm_multi_actgrad_chunks": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _i32, _i32]),
    "cape_spmm_multi_actgrad": (C.c_int, [C.POINTER(CapeSpmmTerm), _i32, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _i64, _i32, _i32, _p, _p]),
    "cape_spmm_multi_actgrad_chunks_bf16": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _i32, _i32]),
    "cape_spmm_multi_actgrad_bf16": (C.c_int, [C.POINTER(CapeSpmmTerm), _i32, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _i64, _i32, _i32, _p, _p]),
    "cape_bwd_prep_spmm_chunks": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _p, _i64, _i32, _i32, _i32, _i32]),
    "cape_spmm_multi_prep_chunks_bf16": (_i32, [C.POINTER(CapeSpmmTerm), _i32, _i32, _i32, _i32]),
    "cape_spmm_multi_prep_bf16": (C.c_int, [C.POINTER(CapeSpmmTerm), _i32, C.c_uint32, _p, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "cape_bwd_prep_spmm_chunks_bf16": (_i32, [_p, _i64, _i32, _p, _i64, _i32, _p, _i64, _i32, _i32, _i32, _i32]),
    "cape_bwd_prep_spmm_bf16": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _i64, _i32, _p, _i64, _i32, _p, _i32, _i32, _i32, _i32, _i32,
                                          _p, _i64, _p, _p, _p]),
    "cape_spmm_multi_prep_chunks": (_i32, [C.POINTER(CapeSpmmTerm), _i32, _i32, _i32, _i32]),
    "cape_spmm_multi_prep": (C.c_int, [C.POINTER(CapeSpmmTerm), _i32, C.c_uint32, _p, _i32, _i32, _i32, _i32, _p, _i64, _p]),
    "cape_bwd_prep_spmm": (C.c_int, [_p, _i64, _i32, _p, _p, _p, _p, _i32, _p, _i64, _i32, _p, _i64, _i32, _p, _i32, _i32, _i32, _i32, _i32,
                                     _p, _i64, _p, _p, _p]),
    "cape_spmm_combine": (C.c_int, [C.POINTER(CapeSpmmTerm), _i32, C.c_uint32, C.POINTER(CapeRank), _p, _i32, _i32, _i32, _p, _p,
                                    _i64, _i32, _i32, _i32, _i32, _p, _p]),
    "cape_bias_act_fwd": (C.c_int, [_p, _i64, _i32, _p, _i32, _i32, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "cape_act_bwd": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _i32, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "cape_colsum_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "cape_colsum": (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _i64, _p]),
    "cape_mask_mul": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "cape_fill_cond": (C.c_int, [_p, _i32, _p, _p, _i64, _i32, _i32, _i32, _i32, _p]),
    "cape_reduce_cond": (C.c_int, [_p, _i64, _i32, _p, _p, _i32, _i32, _i32, _i32, _i32, _p]),
    "cape_groupnorm_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "cape_groupnorm_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _f32, _i32, _i32, _p, _i64, _i32, _p, _p,
                                     _i32, _i32, _i32, _p, _i64, _p, _p]),
    "cape_groupnorm_bwd": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _p, _p, _p, _i32, _i32, _p, _i64, _i32,
                                     _p, _i64, _i32, _p, _p, _p, _i32, _i32, _i32, _p, _i64, _p, _p]),
    "cape_gan_bce_fwd_bwd": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _p, _p, _p, _p, _p, _p]),
    "cape_groupnorm_param_reduce_batch": (C.c_int, [C.c_void_p, _i32, _p]),
    "cape_cond_coef_fwd": (C.c_int, [_p, _i32, _i32, _i32, C.POINTER(CapeCondLayer), _i32, _p]),
    "cape_cond_coef_bwd": (C.c_int, [_p, _i32, _i32, _i32, C.POINTER(CapeCondLayer), _i32, _p, _i32, _i32, _p]),
    "cape_flat_workspace_bytes": (_i64, []),
    "cape_flat_gradnorm": (C.c_int, [_p, _p, _i64, C.POINTER(_i64), _i32, _f32, _f32, _p, _p, _i64, _p]),
    "cape_flat_momentum_update": (C.c_int, [_p, _p, _p, _i64, _f32, _f32, _p, _p, C.POINTER(_i64), _i32, _f32, _f32, _p]),
    "cape_flat_adam_update": (C.c_int, [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _p, _p, _p, C.POINTER(_i64), _i32, _f32, _f32, _p]),
    "cape_sumsq_ranges": (C.c_int, [_p, C.POINTER(_i64), _i32, _f32, _p, _p, _i64, _p]),
    "cape_vae_sample_kl_fwd": (C.c_int, [_p, _p, _p, _p, _i32, _p, _i32, _i32, _p, _i32, _i32, _p]),
    "cape_vae_sample_kl_bwd": (C.c_int, [_p, _p, _p, _p, _i32, _p, _p, _p, _i32, _i32, _p]),
    "cape_fc_long_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "cape_fc_long_fwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), _p, _i64, _p]),
    "cape_fc_long_bwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p), C.POINTER(_p),
                                   _p, _i32, _p]),
    "cape_fc_wide_fwd": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p, _i32, _p, _i32, _p]),
    "cape_fc_wide_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "cape_fc_wide_bwd": (C.c_int, [_p, _i32, _p, _i32, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _i64, _p]),
    "cape_cheb_fused_supported": (C.c_int, [_i32, _i32, _i32]),
    "cape_cheb_fused_debug_timestamps": (C.c_int, [_p]),
    "cape_cheb_fused_fwd": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p]),
    "cape_cheb_fused_bwd_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "cape_cheb_fused_bwd": (C.c_int, [_p, _i64, _i32, _p, _i64, _i32, _p, _p, _i64, _i32, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                      _p, _p, _p, _p, _i32, _p, _i64, _p]),
    "cape_recon_edge_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "cape_recon_edge_loss_fwd_bwd": (C.c_int, [_p, _i32, _p, _p, _p, _p, _p, _i32, _i32, _i32, _f32, _f32,
                                               _p, _p, _p, _f32, _p, _p, _i32, _p, _i64, _p]),
}

# bf16-storage variants with the argument list of their fp32 namesake (include/cape_hip.h, last section)
for _name in ("cape_bwd_prep", "cape_spmm", "cape_spmm_multi", "cape_spmm_combine"):
    SIGNATURES[_name + "_bf16"] = SIGNATURES[_name]
SIGNATURES["cape_colsum_vertex_bf16"] = (C.c_int, [_p, _i64, _i32, _i32, _i32, _i32, _i32, _p, _p])

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

_ERRORS = {-1: "CAPE_EINVAL (bad size / null pointer / unsupported combination)",
           -2: "CAPE_EUNSORTED (CSR columns not strictly increasing)",
           -3: "CAPE_ERANGE (CSR index out of range)",
           -4: "CAPE_EWORKSPACE (workspace too small)"}


class CapeHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise CapeHipError("%s failed: %s" % (what, _ERRORS.get(rc, "hipError_t %d" % rc)))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise CapeHipError("cape_amd needs a HIP device (MI355X / gfx950); no CPU fallback exists")
