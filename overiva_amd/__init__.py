"""overiva_amd -- MI355X (gfx950) implementation of the AuxIVA / OverIVA iteration hot path of
onolab-tmu/overiva behind the reference's own Python signatures.

    from overiva_amd import overiva, auxiva_pca

Host code is Python (as the reference is); all arithmetic on the path runs in hand-written HIP
kernels reached through the C ABI of ``liboveriva_hip.so`` (``include/overiva_hip.h``).  There is
no CPU fallback: without the built library the calls raise ``HipLibraryMissing``.
"""
from . import _lib as _lib_mod

# The host driver of this platform shares device memory between processes through dmabuf only (RCCL, the library's own
# exchanges): the variable must be in the environment BEFORE the process's first GPU call, whatever launched the process.
_lib_mod._set_ipc_env_at_import()

from ._lib import HipError, HipLibraryMissing  # noqa: E402,F401
from .auxiva_pca import auxiva_pca  # noqa: F401
from .ive import ogive  # noqa: F401
from .overiva import (get_device, get_precision, last_solver_info, overiva, release_cached_buffers, set_device,  # noqa: F401
                      set_precision)
from .plan import DeviceX, Plan  # noqa: F401
from .sharded import BinShardedSolver, disable_bin_sharding, enable_bin_sharding, shard_bounds  # noqa: F401

__all__ = ["overiva", "auxiva_pca", "ogive", "Plan", "DeviceX", "BinShardedSolver", "enable_bin_sharding", "disable_bin_sharding",
           "shard_bounds", "release_cached_buffers", "set_device", "get_device", "set_precision", "get_precision", "last_solver_info", "HipError", "HipLibraryMissing"]
