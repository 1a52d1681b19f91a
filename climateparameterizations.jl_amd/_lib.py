"""ctypes binding of libcolnde.so (include/colnde.h).  Loading fails loudly: there is no Python or CPU fallback."""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcolnde.so")
CSRC = os.path.join(_HERE, "csrc")

_F = ctypes.POINTER(ctypes.c_float)
_V = ctypes.c_void_p

# every symbol include/colnde.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("colnde_last_error", ctypes.c_char_p, []),
    ("colnde_version", ctypes.c_int, []),
    ("colnde_min_substeps", ctypes.c_int, [_V]),
    ("colnde_rkc_stages", ctypes.c_int, [_V]),
    ("colnde_create", ctypes.c_int, [_V, ctypes.POINTER(_V)]),
    ("colnde_destroy", None, [_V]),
    ("colnde_n_params", ctypes.c_int, [_V]),
    ("colnde_engine", ctypes.c_int, [_V]),
    ("colnde_set_stream", ctypes.c_int, [_V, _V]),
    ("colnde_set_matrix_arithmetic", ctypes.c_int, [_V, ctypes.c_int]),
    ("colnde_matrix_arithmetic", ctypes.c_int, [_V]),
    ("colnde_set_global_columns", ctypes.c_int, [_V, ctypes.c_int64]),
    ("colnde_set_problem", ctypes.c_int, [_V, _V, _V, _V]),
    ("colnde_rhs", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_forward", ctypes.c_int, [_V, _V, _V]),
    ("colnde_error_estimate", ctypes.c_int, [_V, _V, _F]),
    ("colnde_error_estimate_dev", ctypes.c_int, [_V, _V, _F]),
    ("colnde_choose_substeps", ctypes.c_int, [_V, _V, ctypes.c_float, ctypes.POINTER(ctypes.c_int), _F]),
    ("colnde_substeps", ctypes.c_int, [_V]),
    ("colnde_set_substeps", ctypes.c_int, [_V, ctypes.c_int]),
    ("colnde_flux", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_flux_dev", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_loss_per_tstep", ctypes.c_int, [_V, _V, _V]),
    ("colnde_loss_per_tstep_dev", ctypes.c_int, [_V, _V, _V]),
    ("colnde_infer_dz_wT", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_infer_dz_wT_dev", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_loss", ctypes.c_int, [_V, _V, _F, _F, _F]),
    ("colnde_loss_grad", ctypes.c_int, [_V, _V, _F, _F, _F, _V]),
    ("colnde_infer_forcing", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_set_problem_dev", ctypes.c_int, [_V, _V, _V, _V]),
    ("colnde_rhs_dev", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_forward_dev", ctypes.c_int, [_V, _V, _V]),
    ("colnde_loss_dev", ctypes.c_int, [_V, _V, _F, _V]),
    ("colnde_loss_grad_dev", ctypes.c_int, [_V, _V, _F, _V]),
    ("colnde_infer_forcing_dev", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_convective_adjustment", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, ctypes.c_float, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_convective_adjustment_dev", ctypes.c_int, [_V, _V, _V, _V, ctypes.c_float, ctypes.c_float, ctypes.c_float, _V, ctypes.c_int]),
    ("colnde_implicit_diffusion", ctypes.c_int, [_V, _V, _V, _V, _V, ctypes.c_float, ctypes.c_float, _F, ctypes.c_int, _V, _V, _V, ctypes.c_int]),
    ("colnde_implicit_diffusion_dev", ctypes.c_int, [_V, _V, _V, _V, _V, ctypes.c_float, ctypes.c_float, _F, ctypes.c_int, _V, _V, _V, ctypes.c_int]),
    ("colnde_adam_step_dev", ctypes.c_int, [_V, _V, _V, _V, _V] + [ctypes.c_float] * 6 + [ctypes.c_int]),
    ("colnde_pretrain_flux_dev", ctypes.c_int, [_V, ctypes.c_int, _V, _V, _V, _V, _V, _V, _V, ctypes.c_int] + [ctypes.c_float] * 5 +
     [ctypes.POINTER(ctypes.c_double), ctypes.c_int, _F]),
    ("colnde_coarse_grain_dev", ctypes.c_int, [_V, _V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _V]),
    ("colnde_zscore_stats_dev", ctypes.c_int, [_V, _V, ctypes.c_int64, _V]),
    ("colnde_scale_dev", ctypes.c_int, [_V, _V, ctypes.c_int64, _V, _V]),
    ("colnde_comm_unique_id", ctypes.c_int, [_V]),
    ("colnde_comm_create", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _V, ctypes.c_int, ctypes.POINTER(_V)]),
    ("colnde_comm_destroy", None, [_V]),
    ("colnde_comm_rank", ctypes.c_int, [_V]),
    ("colnde_comm_size", ctypes.c_int, [_V]),
    ("colnde_comm_allreduce_dev", ctypes.c_int, [_V, _V, ctypes.c_int64, ctypes.c_int, _V]),
    ("colnde_allreduce_result_dev", ctypes.c_int, [_V, _V, _V]),
    ("colnde_plan", ctypes.c_int, [_V, ctypes.POINTER(ctypes.c_int)]),
    ("colnde_describe", ctypes.c_int, [_V, ctypes.c_char_p, ctypes.c_int]),
    ("colnde_set_profiling", ctypes.c_int, [_V, ctypes.c_int]),
    ("colnde_kernel_time", ctypes.c_int, [_V, ctypes.c_int, _F, ctypes.POINTER(ctypes.c_int)]),
    ("colnde_reset_kernel_times", ctypes.c_int, [_V]),
]

_lib = None


class ColndeError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libcolnde.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise ColndeError("build did not produce %s" % LIB_PATH)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ColndeError(
                "%s is missing: the HIP extension has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`"
                " or `make -C %s`). colnde has no CPU fallback." % (LIB_PATH, CSRC))
        # libcolnde.so and PyTorch-ROCm both need `libamdhip64.so.7`; the dynamic linker binds that SONAME once per process.
        # PyTorch bundles its own build of the runtime and fails to initialise against the system one, so when torch is
        # installed its copy must be the one that gets loaded: import it first.  (Without torch — e.g. under Julia — the
        # library's RUNPATH finds the system ROCm runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise ColndeError(lib().colnde_last_error().decode("utf-8", "replace"))
