"""ctypes binding of libgpzoo_hip.so (the C ABI declared in include/gpzoo_hip.h).

The library is the product path: there is no CPU or torch fallback.  If the
shared object is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPZ_HIP_LIB") or os.path.join(HERE, "libgpzoo_hip.so")   # override: diagnostic builds only

GPZ_F32, GPZ_F64 = 0, 1
KERNEL_RBF, KERNEL_MATERN32, KERNEL_MGGP_RBF, KERNEL_DISTANCE = 0, 1, 2, 3
SVGP_MATERIALIZE_KZX, SVGP_NARROW_TILES, SVGP_GENERATE_KZX, SVGP_PANEL_PRODUCTS = 1, 2, 4, 8      # gpz_svgp_problem.flags (include/gpzoo_hip.h)
SVGP_BACKWARD_ALGEBRA, SVGP_BACKWARD_CLASSIC = 16, 32
PROF_SLOTS = ("kfill", "stage1", "stage2", "potrf_trailing", "potrf_all", "trtri", "finalize", "_unused")


class KernelDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("n_latent", C.c_int32), ("dtype", C.c_int32), ("n_groups", C.c_int32),
        ("sigma", C.c_void_p), ("lengthscale", C.c_void_p), ("group_a", C.c_void_p), ("group_r2", C.c_void_p),
        ("group_pow", C.c_double),
    ]


class SvgpProblem(C.Structure):
    _fields_ = [
        ("k", KernelDesc),
        ("dtype", C.c_int32), ("whitened", C.c_int32), ("d", C.c_int32), ("flags", C.c_int32),
        ("N", C.c_int64), ("M", C.c_int64),
        ("X", C.c_void_p), ("Z", C.c_void_p), ("gX", C.c_void_p), ("gZ", C.c_void_p),
        ("mu", C.c_void_p), ("Lu_raw", C.c_void_p),
        ("jitter", C.c_double), ("var_clamp_min", C.c_double),
        ("y", C.c_void_p), ("noise_sd", C.c_double),
        ("mean", C.c_void_p), ("scale", C.c_void_p), ("Lu", C.c_void_p), ("chol", C.c_void_p),
        ("kl", C.c_void_p), ("loglik", C.c_void_p), ("elbo", C.c_void_p), ("info", C.c_void_p),
        ("factor_cache", C.c_void_p), ("factor_cache_valid", C.c_int64),
        ("wt_cache", C.c_void_p), ("wt_cache_valid", C.c_int64),
    ]


class SvgpGrads(C.Structure):
    _fields_ = [("g_mean", C.c_void_p), ("g_scale", C.c_void_p), ("scale", C.c_void_p),
                ("grad_mu", C.c_void_p), ("grad_Lu_raw", C.c_void_p),
                ("grad_theta", C.c_void_p), ("grad_Z", C.c_void_p), ("g_chol", C.c_void_p),
                ("g_kl", C.c_void_p), ("point_order", C.c_void_p)]


_SIGNATURES = {
    "gpz_version": (C.c_int, []),
    "gpz_last_error": (C.c_char_p, []),
    "gpz_source_hash": (C.c_char_p, []),
    "gpz_kfill": (C.c_int, [C.POINTER(KernelDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_int32,
                            C.c_void_p]),
    "gpz_kgrad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gpz_kgrad": (C.c_int, [C.POINTER(KernelDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                            C.c_void_p]),
    "gpz_potrf_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gpz_potrf_batched": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "gpz_trsm_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "gpz_trsm_lln_batched": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                       C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gpz_svgp_factor_cache_bytes": (C.c_size_t, [C.POINTER(SvgpProblem)]),
    "gpz_svgp_wt_cache_bytes": (C.c_size_t, [C.POINTER(SvgpProblem), C.c_int64]),
    "gpz_svgp_workspace_bytes": (C.c_size_t, [C.POINTER(SvgpProblem), C.c_int64]),
    "gpz_svgp_forward_path": (C.c_int, [C.POINTER(SvgpProblem), C.c_int64]),
    "gpz_factor_path": (C.c_int, [C.c_int64, C.c_int32]),
    "gpz_svgp_forward": (C.c_int, [C.POINTER(SvgpProblem), C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gpz_svgp_backward_workspace_bytes": (C.c_size_t, [C.POINTER(SvgpProblem), C.c_int64]),
    "gpz_svgp_backward": (C.c_int, [C.POINTER(SvgpProblem), C.POINTER(SvgpGrads), C.c_int64, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "gpz_poisson_nsf_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "gpz_poisson_nsf": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32] +
                        [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p]),
    "gpz_knn": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                          C.c_void_p]),
    "gpz_vnngp_workspace_bytes": (C.c_size_t, [C.POINTER(SvgpProblem), C.c_int32]),
    "gpz_vnngp_state_bytes": (C.c_size_t, [C.POINTER(SvgpProblem)]),
    "gpz_vnngp_forward": (C.c_int, [C.POINTER(SvgpProblem), C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gpz_vnngp_backward_workspace_bytes": (C.c_size_t, [C.POINTER(SvgpProblem), C.c_int32]),
    "gpz_vnngp_backward": (C.c_int, [C.POINTER(SvgpProblem), C.POINTER(SvgpGrads), C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "gpz_wsvgp_precomputed_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "gpz_wsvgp_precomputed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gpz_wsvgp_precomputed_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    "gpz_wsvgp_precomputed_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_int64, C.c_int32] +
                                       [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p]),
    "gpz_comm_unique_id": (C.c_int, [C.c_void_p]),
    "gpz_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "gpz_allreduce_sum_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gpz_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gpz_reduce_scatter_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gpz_allreduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gpz_comm_destroy": (C.c_int, [C.c_void_p]),
    "gpz_profile_enable": (C.c_int, [C.c_int32]),
    "gpz_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int32]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load() -> C.CDLL:
    """Load the HIP library (fails loudly: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    # The HIP runtime must be the one PyTorch ships and initialises (device memory and streams come
    # from torch): import torch first so libamdhip64's SONAME resolves to that copy.  Loading this
    # library before torch binds /opt/rocm's runtime instead and the process then sees no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m gpzoo_amd.build` (hipcc, gfx950). "
            "gpzoo_amd has no CPU/torch fallback for the SVGP hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    check_source_hash(lib)
    _lib = lib
    return lib


def check_source_hash(lib) -> None:
    """Refuse a binary that was not built from the sources lying next to it: file times do not survive a copy, so the
    library carries the sha256 of its sources (gpz_source_hash) and it is compared by content.  No sources on disk (a
    binary-only deployment) -> nothing to compare."""
    from . import build
    want = build.source_hash()
    have = lib.gpz_source_hash().decode("ascii", "replace")
    if want is not None and have != want:
        raise RuntimeError(
            f"{LIB_PATH} was built from other sources (binary {have}, gpzoo_amd/csrc {want}): "
            "rebuild it with `python -m gpzoo_amd.build`")


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gpz_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")
