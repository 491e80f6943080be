"""ctypes binding of libvqhip.so (include/vqhip.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VQHIP_LIB: another build of the same ABI (A/B timing of two builds on one box); default: the in-tree library
LIB_PATH = os.environ.get("VQHIP_LIB") or os.path.join(_HERE, "libvqhip.so")

OK, ERR_INVALID, ERR_STATE, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
KERNEL_AUTO, KERNEL_SCALAR, KERNEL_MFMA = 0, 1, 2
PROJECT_OMEGA, PROJECT_LAMBDA = 0, 1
OPT_RANKING, OPT_PROJECTION = 0, 1
VALUE_DEFAULT, VALUE_F32, VALUE_BF16 = 0, 1, 2

# name -> (restype, argtypes); the single list the symbol-export test walks too
_f, _u, _v = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
SIGNATURES = {
    "vqhip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "vqhip_destroy": (C.c_int, [_v]),
    "vqhip_last_error": (C.c_char_p, [_v]),
    "vqhip_set_prototypes": (C.c_int, [_v, _f, C.c_uint32, C.c_uint32]),
    "vqhip_set_labels": (C.c_int, [_v, _u, C.c_uint32]),
    "vqhip_set_omega": (C.c_int, [_v, _f, C.c_uint32, C.c_uint32]),
    "vqhip_set_local_omegas": (C.c_int, [_v, _f, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vqhip_set_kernel_policy": (C.c_int, [_v, C.c_int]),
    "vqhip_set_option": (C.c_int, [_v, C.c_int, C.c_int]),
    "vqhip_assign": (C.c_int, [_v, _f, C.c_uint64, _u, _f]),
    "vqhip_assign_rows_f64": (C.c_int, [_v, C.POINTER(C.POINTER(C.c_double)), C.c_uint64, C.c_uint32, _u, _f]),
    "vqhip_project": (C.c_int, [_v, _f, C.c_uint64, C.c_int, _f]),
    "vqhip_project_device": (C.c_int, [_v, _v, C.c_uint64, C.c_int, _v, _v]),
    "vqhip_project_rows_f64": (C.c_int, [_v, _v, C.c_uint64, C.c_uint32, C.c_int, _v]),
    "vqhip_set_prototypes_f64": (C.c_int, [_v, C.POINTER(C.c_double), C.c_uint32, C.c_uint32]),
    "vqhip_assign_f64": (C.c_int, [_v, C.POINTER(C.c_double), C.c_uint64, _u, C.POINTER(C.c_double)]),
    "vqhip_assign_matched_f64": (C.c_int, [_v, C.POINTER(C.c_double), _u, C.c_uint64, _u, C.POINTER(C.c_double), _u,
                                           C.POINTER(C.c_double)]),
    "vqhip_assign_device": (C.c_int, [_v, _v, C.c_uint64, _v, _v, _v]),
    "vqhip_prepare": (C.c_int, [_v]),
    "vqhip_assign_device_noalloc": (C.c_int, [_v, _v, C.c_uint64, _v, _v, _v]),
    "vqhip_assign_matched": (C.c_int, [_v, _f, _u, C.c_uint64, _u, _f, _u, _f]),
    "vqhip_assign_matched_rows_f64": (C.c_int, [_v, C.POINTER(C.POINTER(C.c_double)), _u, C.c_uint64, C.c_uint32, _u, _f, _u, _f]),
    "vqhip_assign_matched_device": (C.c_int, [_v, _v, _v, C.c_uint64, _v, _v, _v, _v, _v]),
    "vqhip_glvq_batch_gradient": (C.c_int, [_v, _f, _u, C.c_uint64, _f, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "vqhip_glvq_batch_gradient_device": (C.c_int, [_v, _v, _v, C.c_uint64, _v, _v, _v]),
    "vqhip_multi_glvq_batch_gradient": (C.c_int, [_v, _f, _u, C.c_uint64, _f, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "vqhip_last_kernel": (C.c_char_p, [_v]),
    "vqhip_last_ranking": (C.c_char_p, [_v]),
    "vqhip_last_projection": (C.c_char_p, [_v]),
    "vqhip_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "vqhip_multi_destroy": (C.c_int, [_v]),
    "vqhip_multi_last_error": (C.c_char_p, [_v]),
    "vqhip_multi_device_count": (C.c_int, [_v]),
    "vqhip_multi_set_prototypes": (C.c_int, [_v, _f, C.c_uint32, C.c_uint32]),
    "vqhip_multi_set_labels": (C.c_int, [_v, _u, C.c_uint32]),
    "vqhip_multi_set_omega": (C.c_int, [_v, _f, C.c_uint32, C.c_uint32]),
    "vqhip_multi_set_local_omegas": (C.c_int, [_v, _f, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vqhip_multi_set_option": (C.c_int, [_v, C.c_int, C.c_int]),
    "vqhip_multi_assign": (C.c_int, [_v, _f, C.c_uint64, _u, _f]),
    "vqhip_multi_assign_device": (C.c_int, [_v, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "vqhip_multi_assign_rows_f64": (C.c_int, [_v, C.POINTER(C.POINTER(C.c_double)), C.c_uint64, C.c_uint32, _u, _f]),
    "vqhip_multi_assign_matched": (C.c_int, [_v, _f, _u, C.c_uint64, _u, _f, _u, _f]),
    "vqhip_synth_fill_host": (C.c_int, [_f, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_float, _f, C.c_uint64]),
    "vqhip_synth_fill_device": (C.c_int, [_v, _v, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_float, _v,
                                          C.c_uint64, _v]),
    "vqhip_version": (C.c_char_p, []),
}

_lib = None


def _share_torch_hip_runtime():
    """The PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64 (soname libamdhip64.so.7, the
    same as /opt/rocm's).  Two HIP runtimes in one process cannot both own the GPU, and a
    hipStream_t of one means nothing to the other.  Loading torch's copy first makes the dynamic
    loader satisfy libvqhip's NEEDED libamdhip64.so.7 with it, so torch tensors, torch streams
    and this library share one runtime.  Without torch installed the system ROCm runtime is used."""
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Loads vq_amd/libvqhip.so (built by `make` / __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc, gfx950). "
                "vq_amd has no CPU or PyTorch fallback.")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


class VqhipError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"vqhip error {code}: {text}")
        self.code = code
        self.text = text
