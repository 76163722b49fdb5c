"""ctypes binding of libsmoe_hip.so (C ABI: include/smoe_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of the engine fails loudly (build it with ``python -c 'import __graft_entry__ as g;
g.build()'`` or ``make -C steered_mixture_of_experts_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMOE_HIP_LIBRARY: another build of the same library (the SMOE_DEBUG one: `make -C csrc debug`), never a different backend
LIB_PATH = os.environ.get("SMOE_HIP_LIBRARY") or os.path.join(_HERE, "libsmoe_hip.so")

SMOE_ABI_VERSION = 2
SMOE_OK = 0
SMOE_ERR_INVALID = -1
SMOE_ERR_UNSUPPORTED = -2
SMOE_ERR_HIP = -3
SMOE_ERR_NO_DEVICE = -4

EXPORTS = (
    "smoe_create", "smoe_destroy", "smoe_is_supported", "smoe_padded_kernels", "smoe_get_coords", "smoe_forward",
    "smoe_fit", "smoe_update_kernel_list", "smoe_checkpoint_best", "smoe_reduce_scalars",
    "smoe_fit_variant", "smoe_fit_occupancy", "smoe_set_tiling", "smoe_last_error", "smoe_abi_version",
    "smoe_shared_create", "smoe_shared_destroy", "smoe_shared_num_batches", "smoe_shared_list_words",
    "smoe_shared_forward", "smoe_shared_accumulate", "smoe_shared_apply", "smoe_shared_grad_buffer",
    "smoe_shared_fit", "smoe_shared_update_kernel_list", "smoe_shared_set_loss_weights",
    "smoe_set_center_grid", "smoe_shared_set_center_grid", "smoe_set_total_blocks", "smoe_padded_kernels_full", "smoe_shared_discard", "smoe_set_sampling",
)


class SmoeConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("dim", C.c_int32),
        ("block_shape", C.c_int32 * 3), ("channels", C.c_int32), ("kernels", C.c_int32),
        ("precision", C.c_int32), ("margin", C.c_float), ("use_determinant", C.c_int32),
        ("use_yuv", C.c_int32), ("train_pis", C.c_int32), ("train_gammas", C.c_int32),
        ("train_musx", C.c_int32), ("lr_expert", C.c_float), ("lr_pis", C.c_float),
        ("lr_steer", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
        ("adam_eps", C.c_float), ("grad_clip", C.c_float), ("pis_l1", C.c_float),
        ("u_l1", C.c_float), ("start_pis", C.c_int32), ("only_y_gamma", C.c_int32), ("ssim_opt", C.c_int32),
        ("quantization_mode", C.c_int32), ("quantize_pis", C.c_int32), ("bit_depths", C.c_int32 * 5),
        ("lower_bounds", C.c_float * 5), ("upper_bounds", C.c_float * 5), ("train_inverse_cov", C.c_int32),
        ("radial_as", C.c_int32), ("kernel_count_as_norm_l1", C.c_int32),
    ]


class SmoeSharedConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("dim", C.c_int32),
        ("image_shape", C.c_int32 * 3), ("batch_shape", C.c_int32 * 3),
        ("channels", C.c_int32), ("kernels", C.c_int32), ("precision", C.c_int32), ("margin", C.c_float),
        ("use_determinant", C.c_int32), ("use_yuv", C.c_int32), ("train_pis", C.c_int32),
        ("train_gammas", C.c_int32), ("train_musx", C.c_int32), ("lr_expert", C.c_float), ("lr_pis", C.c_float),
        ("lr_steer", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("grad_clip", C.c_float), ("pis_l1", C.c_float), ("u_l1", C.c_float), ("start_pis", C.c_int32),
        ("only_y_gamma", C.c_int32), ("overlap", C.c_int32),
        ("quantization_mode", C.c_int32), ("quantize_pis", C.c_int32), ("bit_depths", C.c_int32 * 5),
        ("lower_bounds", C.c_float * 5), ("upper_bounds", C.c_float * 5), ("ssim_opt", C.c_int32),
        ("train_inverse_cov", C.c_int32), ("radial_as", C.c_int32), ("kernel_count_as_norm_l1", C.c_int32),
    ]


class SmoeParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("pis", "musX", "A_diagonal", "A_corr", "gamma_e", "nu_e")]


class SmoeAdamState(C.Structure):
    _fields_ = [("m", SmoeParams), ("v", SmoeParams), ("beta1_power", C.c_float),
                ("beta2_power", C.c_float), ("step", C.c_int64)]


class SmoeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsmoe_hip error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libsmoe_hip.so once and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built.  This package has no CPU "
            "path; run __graft_entry__.build() (hipcc --offload-arch=gfx950).")
    lib = C.CDLL(LIB_PATH)
    vp, i32, fp = C.c_void_p, C.c_int32, C.c_void_p
    lib.smoe_create.argtypes = [C.POINTER(vp), C.POINTER(SmoeConfig)]
    lib.smoe_destroy.argtypes = [vp]
    lib.smoe_is_supported.argtypes = [i32, i32, i32]
    lib.smoe_padded_kernels.argtypes = [i32, i32, i32]
    lib.smoe_padded_kernels_full.argtypes = [i32, i32, i32]
    lib.smoe_get_coords.argtypes = [vp, fp]
    lib.smoe_forward.argtypes = [vp, i32, fp, fp, C.POINTER(SmoeParams), fp, fp, fp, fp, fp, fp, i32, vp]
    lib.smoe_fit.argtypes = [vp, i32, fp, fp, C.POINTER(SmoeParams), C.POINTER(SmoeAdamState), i32,
                             fp, fp, fp, fp, fp, vp]
    lib.smoe_update_kernel_list.argtypes = [vp, i32, C.POINTER(SmoeParams), fp, vp]
    lib.smoe_checkpoint_best.argtypes = [vp, i32, fp, fp, C.POINTER(SmoeParams), C.POINTER(SmoeParams), vp]
    lib.smoe_reduce_scalars.argtypes = [vp, i32, fp, fp, fp, fp, vp]
    lib.smoe_fit_variant.argtypes = [vp, i32]
    lib.smoe_fit_variant.restype = C.c_char_p
    lib.smoe_fit_occupancy.argtypes = [vp, i32]
    lib.smoe_set_tiling.argtypes = [vp, i32]
    lib.smoe_set_total_blocks.argtypes = [vp, C.c_int64]
    lib.smoe_set_sampling.argtypes = [vp, i32]
    lib.smoe_last_error.restype = C.c_char_p
    lib.smoe_shared_create.argtypes = [C.POINTER(vp), C.POINTER(SmoeSharedConfig)]
    lib.smoe_shared_destroy.argtypes = [vp]
    lib.smoe_shared_num_batches.argtypes = [vp]
    lib.smoe_shared_list_words.argtypes = [vp]
    lib.smoe_shared_forward.argtypes = [vp, i32, i32, fp, C.POINTER(SmoeParams), fp, fp, fp, fp, fp, i32, vp]
    lib.smoe_shared_accumulate.argtypes = [vp, i32, i32, fp, C.POINTER(SmoeParams), fp, fp, fp, vp]
    lib.smoe_shared_apply.argtypes = [vp, C.POINTER(SmoeParams), C.POINTER(SmoeAdamState), vp]
    lib.smoe_shared_grad_buffer.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.smoe_shared_fit.argtypes = [vp, fp, C.POINTER(SmoeParams), C.POINTER(SmoeAdamState), i32, fp, fp, fp, vp]
    lib.smoe_shared_update_kernel_list.argtypes = [vp, i32, i32, C.POINTER(SmoeParams), fp, vp]
    lib.smoe_shared_set_loss_weights.argtypes = [vp, fp]
    lib.smoe_shared_discard.argtypes = [vp, vp]
    lib.smoe_set_center_grid.argtypes = [vp, fp]
    lib.smoe_shared_set_center_grid.argtypes = [vp, fp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("smoe_fit_variant", "smoe_last_error"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != SMOE_OK:
        raise SmoeError(code, load().smoe_last_error().decode())
