"""ctypes binding of liboveriva_hip.so (C ABI declared in include/overiva_hip.h).

There is no fallback: if the shared library is missing or cannot be loaded the import of the
product path fails with an explicit error.  ``python -m overiva_amd.build`` (or
``__graft_entry__.build()``) produces the library.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# ($OIVA_LIB: another build of the same library -- kernel variants under measurement, tools/build_variant.py)
LIB_PATH = os.environ.get("OIVA_LIB") or os.path.join(_PKG, "liboveriva_hip.so")

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_NUMERIC = 0, -1, -2, -3, -4
MODEL_IDS = {"laplace": 0, "gauss": 1}
N_STAGES = 4
# oiva_plan_set_precision flags (include/overiva_hip.h)
PREC_FAST, PREC_UPDATE_F64, PREC_UPDATE_ROWS, PREC_COV_F64 = 0, 1, 2, 4
PREC_PRECISE = PREC_UPDATE_F64 | PREC_COV_F64
PREC_MIXED = PREC_UPDATE_F64
PREC_BY_NAME = {"fast": PREC_FAST, "mixed": PREC_MIXED, "precise": PREC_PRECISE}
STAGE_NAMES = ("demix_power", "activation", "weighted_cov", "ip_update")


class HipLibraryMissing(ImportError):
    pass


class HipError(RuntimeError):
    """a HIP runtime call failed inside the library"""


_lib = None

_vp, _i, _ll, _fp = C.c_void_p, C.c_int, C.c_longlong, C.POINTER(C.c_float)

# name -> argtypes ; every function returns int (see include/overiva_hip.h)
SIGNATURES = {
    "oiva_device_count": [C.POINTER(_i)],
    "oiva_plan_create": [C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i, _vp],
    "oiva_plan_destroy": [_vp],
    "oiva_plan_set_x_host": [_vp, _vp, _ll],
    "oiva_plan_set_x_dev": [_vp, _vp],
    "oiva_plan_set_x_host_c128": [_vp, _vp, _ll],
    "oiva_plan_covariance": [_vp],
    "oiva_plan_get_cx": [_vp, _vp, _i],
    "oiva_plan_set_w": [_vp, _vp, _i],
    "oiva_plan_iterate": [_vp, _i],
    "oiva_plan_power": [_vp],
    "oiva_plan_power_buffer": [_vp, _i, C.POINTER(_vp), C.POINTER(_ll)],
    "oiva_plan_update": [_vp, _vp, _i],
    "oiva_plan_demix": [_vp, _vp, _ll, _i],
    "oiva_plan_demix_dev": [_vp, _i, _vp],
    "oiva_plan_set_io_slab": [_vp, _ll],
    "oiva_host_prefault": [_vp, _ll],
    "oiva_pool_trim": [],
    "oiva_plan_demix_c128": [_vp, _vp, _ll, _i],
    "oiva_plan_set_w_pca": [_vp, _vp],
    "oiva_plan_set_w_eig": [_vp],
    "oiva_xchg_create": [_vp, _i, _i, _i, _ll],
    "oiva_xchg_export": [_vp, _vp],
    "oiva_xchg_connect": [_vp, _vp],
    "oiva_xchg_push": [_vp, _vp, _vp, _ll, _i],
    "oiva_xchg_wait": [_vp, _vp, _i],
    "oiva_xchg_gathered": [_vp, _i, _vp],
    "oiva_xchg_poll": [_vp, _i, _i, _vp],
    "oiva_xchg_force": [_vp, _i],
    "oiva_xchg_destroy": [_vp],
    "oiva_plan_get_w": [_vp, _vp, _i],
    "oiva_plan_sync": [_vp],
    "oiva_plan_save_w": [_vp],
    "oiva_plan_restore_w": [_vp],
    "oiva_plan_iterate_timed": [_vp, _i, _fp, _fp],
    "oiva_plan_get_cov_splits": [_vp, C.POINTER(_i)],
    "oiva_plan_set_cov_splits": [_vp, _i],
    "oiva_plan_set_cov_quad": [_vp, _i, C.POINTER(_i)],
    "oiva_plan_set_pow_splits": [_vp, _i],
    "oiva_plan_set_cov_hmfma": [_vp, _i],
    "oiva_plan_set_fuse_cov_update": [_vp, _i, C.POINTER(_i)],
    "oiva_plan_use_graph": [_vp, _i],
    "oiva_plan_set_precision": [_vp, _i],
    "oiva_plan_set_resident": [_vp, _i],
    "oiva_plan_resident_info": [_vp, C.POINTER(_i)],
    "oiva_plan_resident_phases": [_vp, C.POINTER(C.c_double), C.POINTER(_i)],
    "oiva_plan_resident_debug": [_vp, _i, _i],
    "oiva_plan_resident_debug_from": [_vp, _i, _i, _i],
    "oiva_plan_resident_loopback": [_vp, _i],
    "oiva_plan_fused_connect": [_vp, _vp],
    "oiva_plan_fused_loopback": [_vp, _i],
    "oiva_plan_fused_debug": [_vp, _i, _i],
    "oiva_plan_resident_connect": [_vp, _vp],
    "oiva_plan_resident_trace": [_vp, _i, _vp, C.POINTER(_i), C.POINTER(_i)],
    "oiva_plan_set_resident_splits": [_vp, _i],
    "oiva_test_set_rinv": [_vp, _vp],
    "oiva_test_get_rinv": [_vp, _vp, _vp],
    "oiva_test_run_weighted_cov": [_vp],
    "oiva_test_get_v": [_vp, _vp, _i],
    "oiva_test_run_update": [_vp],
    "oiva_test_get_what": [_vp, _vp, _i],
    "oiva_test_set_what": [_vp, _vp, _i],
    "oiva_test_run_power": [_vp, _vp],
    "oiva_test_time_stage": [_vp, _i, _i, _fp],
    "oiva_plan_ogive_begin": [_vp, _i, _i],
    "oiva_plan_ogive_iterate": [_vp, _i, _i, C.c_double, C.c_double, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_double)],
    "oiva_stft_create": [C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp],
    "oiva_stft_destroy": [_vp],
    "oiva_stft_shape": [_vp, C.POINTER(_i), C.POINTER(_i)],
    "oiva_stft_analysis": [_vp, _vp, _vp, C.POINTER(_vp)],
    "oiva_stft_synthesis": [_vp, _vp, _i, _vp],
}


def load():
    """Load the library once; raise HipLibraryMissing if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own copy of the HIP runtime.  Two HIP runtimes in one process do not share
    # the GPU: whichever initialises second reports "No HIP GPUs are available".  When torch is installed (the
    # multi-GPU driver, bench.py and the tests use it for process groups and device tensors) it is imported
    # first, so that its runtime is the one already loaded when liboveriva_hip.so resolves libamdhip64.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run `python -m overiva_amd.build` "
            "(needs hipcc).  overiva_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryMissing(f"cannot load {LIB_PATH}: {e}") from e
    lib.oiva_version.restype = C.c_int
    lib.oiva_version.argtypes = []
    lib.oiva_last_error.restype = C.c_char_p
    lib.oiva_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = lib
    return lib


def require_dmabuf_ipc(what):
    """The host driver of this platform exports device memory through dmabuf only: ``hipIpcGetMemHandle`` (the push
    exchange and the X-resident kernel's exchange between processes, RCCL, device tensors shared across processes) fails
    with "invalid argument" unless ``HSA_ENABLE_IPC_MODE_LEGACY=0`` is in the environment BEFORE the process's first GPU
    call.  ``import overiva_amd`` sets the variable when it is unset (``__init__.py``: importing torch does not start the
    GPU runtime, so a process that imports the package before its first GPU call is safe under ANY launcher); the
    multi-process exchanges call this when they are asked for and warn when the runtime had started without it, or when
    the variable holds another value.  Returns None when all is well, else the reason (``exchange_degraded`` of bench.py)."""
    import warnings

    cur = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    why = None
    if cur is None or _IPC_ENV_SET_LATE:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        try:
            import torch

            started = torch.cuda.is_initialized()
        except Exception:
            started = False
        if (started and cur is None) or _IPC_ENV_SET_LATE:
            why = (f"{what}: HSA_ENABLE_IPC_MODE_LEGACY=0 was not in the environment when the GPU runtime started; "
                   "exporting device memory to other processes may fail (import overiva_amd, or export the variable, before "
                   "the first GPU call)")
    elif cur != "0":
        why = f"{what}: HSA_ENABLE_IPC_MODE_LEGACY={cur!r}; this platform's driver supports dmabuf IPC (=0) only"
    if why:
        warnings.warn(why)
    return why


_IPC_ENV_SET_LATE = False


def _set_ipc_env_at_import():
    """``import overiva_amd``: HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller chose a value; remembers when the GPU runtime of
    this process was already running (the variable is read once, at its start)"""
    global _IPC_ENV_SET_LATE
    if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
        return
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import sys

    t = sys.modules.get("torch")
    try:
        _IPC_ENV_SET_LATE = bool(t is not None and t.cuda.is_initialized())
    except Exception:
        _IPC_ENV_SET_LATE = False


def check(rc):
    if rc == OK:
        return
    msg = load().oiva_last_error().decode("utf-8", "replace")
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_NUMERIC:
        raise np.linalg.LinAlgError(msg)
    if rc == ERR_STATE:
        raise RuntimeError(msg)
    raise HipError(msg)


def ptr(a):
    """raw pointer of a C-contiguous numpy array"""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)
