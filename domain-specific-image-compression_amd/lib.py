"""ctypes binding of libdsic_hip.so (include/dsic_hip.h).

The product path has no CPU fallback: `load()` raises if the library has not
been built, and every wrapper raises RuntimeError/ValueError on a non-zero
status with the library's own message.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int, c_int64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdsic_hip.so")

DSIC_OK, DSIC_EINVAL, DSIC_EHIP = 0, 1, 2
ACT_NONE, ACT_GDN, ACT_IGDN, ACT_RELU = 0, 1, 2, 3

_P = c_void_p
# name -> (restype, argtypes); mirrors include/dsic_hip.h one to one
SIGNATURES = {
    "dsic_last_error": (ctypes.c_char_p, []),
    "dsic_abi_version": (c_int, []),
    "dsic_split_bf16": (c_int, []),
    "dsic_set_split_bf16": (c_int, [c_int]),
    "dsic_packed_conv_weight_floats": (c_int64, [c_int, c_int, c_int]),
    "dsic_pack_conv_weight": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "dsic_pack_convT_weight": (c_int, [_P, _P, c_int, c_int, _P]),
    "dsic_convT_image_weight_floats": (c_int64, [c_int]),
    "dsic_pack_convT_image_weight": (c_int, [_P, _P, c_int, c_int, _P]),
    "dsic_image_to_nhwc8": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dsic_nhwc_to_nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dsic_nchw_to_nhwc": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dsic_reflect_pad_br": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dsic_normalize_bands": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "dsic_conv2d_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_int, _P]),
    "dsic_wino_weight_floats": (c_int64, [c_int, c_int]),
    "dsic_pack_wino_weight": (c_int, [_P, _P, c_int, c_int, _P]),
    "dsic_conv3x3_wino_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       _P, _P]),
    "dsic_pack_wino_s2_weight": (c_int, [_P, _P, c_int, c_int, _P]),
    "dsic_pack_wino_convT_weight": (c_int, [_P, _P, c_int, c_int, _P]),
    "dsic_conv_transpose2d_wino_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                                _P]),
    "dsic_wino_bf16_planes": (c_int, []),
    "dsic_wino_bf16_weight_bytes": (c_int64, [c_int, c_int]),
    "dsic_split_wino_weight_bf16": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "dsic_conv3x3_wino_bf16_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_int, _P, _P]),
    "dsic_wino_bf16_ksplit": (c_int, [c_int, c_int, c_int]),
    "dsic_wino_bf16_m64": (c_int, [c_int, c_int, c_int, c_int]),
    "dsic_conv3x3_wino_bf16_splitk_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                                   c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "dsic_conv_transpose2d_wino_bf16_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                                     _P, _P]),
    "dsic_conv_first_nchw": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dsic_conv_first_u8hwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dsic_image_u8hwc_to_f32nchw": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dsic_conv_transpose2d_nhwc": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int,
                                           c_int, c_int, _P]),
    "dsic_conv_transpose2d_image": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "dsic_hyper_params": (c_int, [_P] * 13 + [c_int, c_int, c_int, c_int, ctypes.c_float,
                                     ctypes.c_float, _P]),
    "dsic_rate_workspace_doubles": (c_int64, [c_int]),
    "dsic_rate": (c_int, [_P] * 14 + [c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "dsic_sigma_nu_spatial": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, ctypes.c_float, ctypes.c_float, _P]),
    "dsic_student_t_bits": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "dsic_gaussian_bits": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "dsic_round": (c_int, [_P, _P, c_int64, _P]),
    "dsic_gdn_nchw": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dsic_ssim_partial_doubles": (c_int64, [c_int, c_int, c_int]),
    "dsic_ssim_level": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, ctypes.c_float, ctypes.c_float,
                                c_int, _P]),
    "dsic_ssim_level_pool_fused": (c_int, [c_int, c_int]),
    "dsic_ssim_level_pool": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, ctypes.c_float, ctypes.c_float,
                                     c_int, _P]),
    "dsic_avgpool2": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "dsic_msssim_finalize": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "dsic_sqerr_per_image": (c_int, [_P, _P, _P, c_int, c_int64, c_int, _P]),
    "dsic_latent_support": (c_int, [_P, _P, _P, c_int, c_int64, c_int64, c_int, _P]),
    "dsic_cdf_tables_gauss": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "dsic_cdf_tables_student": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "dsic_range_encode": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                  c_int64, c_int64, _P, _P, c_int, c_int, _P]),
    "dsic_stream_create_masked": (c_int, [_P, c_int, _P]),
    "dsic_stream_destroy": (c_int, [_P]),
    "dsic_range_decode": (c_int, [_P, c_int64, _P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int,
                                  c_int, c_int, _P, _P, _P]),
    "dsic_host_normal_cdf": (ctypes.c_double, [ctypes.c_double]),
    "dsic_host_student_t_cdf": (ctypes.c_double, [ctypes.c_double, ctypes.c_double]),
    "dsic_host_gaussian_cdf_f32": (ctypes.c_float, [ctypes.c_float]),
    "dsic_host_exp_f32": (ctypes.c_float, [ctypes.c_float]),
    "dsic_host_pmf_to_uint16_cdf": (c_int, [_P, c_int, c_int, _P]),
    "dsic_host_cdf_table": (c_int, [c_int, ctypes.c_float, ctypes.c_float, c_int, c_int, _P, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load libdsic_hip.so; never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with "
                "`python domain-specific-image-compression_amd/build.py` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); import it
        # first so this library binds to the SAME runtime instance that owns the
        # streams and allocations (loading /opt/rocm's copy beside it gives
        # "no ROCm-capable device").
        import torch  # noqa: F401
        # A rank drives up to six HIP streams (main, hyperprior branch, two coder streams, RCCL, copies); on the HIP
        # runtime's default of 4 hardware queues two of them share one and wait for each other (bench: -20 % under
        # RCCL).  The setting is read when the runtime initialises, so it is made here, before the first HIP call of
        # a process that has not made one yet; a process that already has is told.
        if "GPU_MAX_HW_QUEUES" not in os.environ:
            if torch.cuda.is_initialized():
                import warnings
                warnings.warn("dsic_amd: the HIP runtime was initialised without GPU_MAX_HW_QUEUES=8; the coder's side "
                              "streams will share hardware queues with the conv stream (export it before starting python)")
            else:
                os.environ["GPU_MAX_HW_QUEUES"] = "8"
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status == DSIC_OK:
        return
    msg = load().dsic_last_error().decode(errors="replace")
    if status == DSIC_EINVAL:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")
