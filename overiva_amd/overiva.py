"""Drop-in ``overiva()`` -- same name, argument order, defaults and return values as reference
``overiva.py:28-38``; the body runs on one MI355X (or bin-sharded over several, see ``sharded.py``).

Arithmetic (``set_precision``): device data (X, Y) is complex64 whatever the input dtype.  ``"auto"`` (default) follows
the reference, which computes in the dtype of X (``overiva.py:89,126-131``): complex128 input runs ``"precise"`` -- the
weighted covariance as float64 sums of exact float64 products and the per-bin algebra in float64 with W_hat carried in
complex128, i.e. the reference's complex128 arithmetic applied to complex64-rounded data; complex64 input runs the same on
short frame axes with up to 8 channels (the reference's own call sizes: it forms the covariances in complex128 there too,
``overiva.py:179``) and ``"mixed"`` elsewhere -- float32 products and short float32 lane chains in the covariance pass, every
longer sum and the per-bin algebra in float64: closer to the reference's complex128 result than the reference's own complex64
arithmetic is (``resolve_precision``).  ``"fast"`` is float32 in the per-bin algebra too:
1e-5 on well-conditioned input, a few times the reference's complex64 noise otherwise
(tests/test_gpu_parity.py::test_fast_mode_accuracy).

Documented deviations from the reference:
* X is held as complex64 on the device even for complex128 input (a 6e-8 relative input perturbation); the
  result is cast back to the input's complex dtype;
* at most 16 channels (``OIVA_MAX_CHANNELS``; the reference has no limit);
* an unknown ``model`` raises ``ValueError`` (the reference silently returns NaN, ``overiva.py:152-167``);
* the returned ``W`` is a fresh contiguous ``(n_freq, n_chan, n_src)`` array, not a view of ``W_hat``
  (``overiva.py:90,201-202``).
"""
import atexit
import os

import numpy as np

from . import sharded
from ._lib import HipError
from .plan import DeviceX, Plan

_device = None
_precision = "auto"
MODES = ("auto", "fast", "mixed", "precise")


def set_precision(mode):
    """Arithmetic of subsequent ``overiva()`` / ``auxiva_pca()`` calls in this process: ``"auto"`` (default),
    ``"fast"``, ``"mixed"`` or ``"precise"`` (see the module docstring)."""
    global _precision
    if mode not in MODES:
        raise ValueError(f"precision must be one of {MODES}")
    _precision = mode


# frame axes up to this long count as short: the reference's own calls (4096-point frames of ~10 s of audio: 160-235 frames)
SHORT_FRAME_AXIS = 256


def resolve_precision(dtype, n_chan, mode=None, n_src=None, n_frames=None):
    """``"auto"`` follows the reference, which computes in the dtype of X (overiva.py:89,126,131): complex128 input ->
    ``"precise"``.  complex64 input: the reference still forms every weighted covariance in complex128 (``r_inv`` is float64,
    overiva.py:127-128,179) and rounds it once --

    * up to 8 channels on a short frame axis (``n_frames <= SHORT_FRAME_AXIS``, every call of the reference's own scripts): the
      same, ``"precise"`` -- float64 sums of exact float64 products, float64 per-bin algebra.  Those passes are bound by
      latency, not arithmetic: 0-3 % of the iteration up to 5 channels, 3-18 % at 6-8 channels with up to 4 sources
      (profiles/r05_stage_times_reference_shapes.log), and it removes the one i.i.d. row that float32 chains held 1.7 reference
      floors from the complex128 result (round 5's ``NOISE_ROWS_OVER_ONE_FLOOR``).  (2, 6 and 8 channels with 1-2 sources whose
      X fits on chip keep ``"mixed"`` and the X-resident kernel -- the float64 covariance exists there for 4 channels only --,
      decided by ``_SingleDevice`` once the plan knows its geometry);
    * everything else -> ``"mixed"``: float32 products and short float32 lane chains in the covariance pass, every longer sum
      and the per-bin algebra in float64 (up to 8 channels: csrc/kernels_cov.hip, kernels_cov_pair32.hip; 9..16 channels:
      kernels_cov_quad.hip with up to 4 sources, kernels_cov_half16.hip with 5..8, the fp32 matrix cores of
      kernels_cov_hmfma.hip with 9..16 -- odd counts on a copy of X padded by one zero channel).
    (``n_src`` does not matter; kept for callers.)"""
    mode = _precision if mode is None else mode
    if mode != "auto":
        return mode
    if np.dtype(dtype) != np.complex64:
        return "precise"
    if n_frames is not None and n_frames <= SHORT_FRAME_AXIS and n_chan <= 8:
        return "precise"
    return "mixed"


def get_precision():
    return _precision


def release_cached_buffers():
    """give the large device buffers the library keeps between calls (copies of X and Y of destroyed plans, at most
    $OIVA_POOL_MB = 2048 MB) back to the driver"""
    from . import _lib

    while _plan_cache:
        _plan_cache.pop(next(iter(_plan_cache))).close()
    _lib.check(_lib.load().oiva_pool_trim())


def _release_at_exit():
    """kept plans are destroyed while the HIP runtime is still there, not by ``Plan.__del__`` at interpreter finalisation"""
    from . import _lib

    if _lib._lib is None or not _plan_cache:      # (nothing kept: no GPU call at exit -- the process's memory goes with it anyway)
        return
    try:
        while _plan_cache:
            _plan_cache.pop(next(iter(_plan_cache))).close()
    except Exception:
        pass


atexit.register(_release_at_exit)


_last_info = {}


def last_solver_info():
    """what the last ``overiva()`` call of this process ran on: arithmetic mode, whether the X-resident kernel was used
    (launches) and whether it had to fall back to the four-launch path"""
    return dict(_last_info)


def set_device(index):
    """GPU used by subsequent calls in this process (default: $LOCAL_RANK, else 0)."""
    global _device
    _device = int(index)


def get_device():
    if _device is not None:
        return _device
    return int(os.environ.get("LOCAL_RANK", "0"))


def _complex_dtype(X):
    if X.dtype == np.complex64 or X.dtype == np.complex128:
        return X.dtype
    if np.issubdtype(X.dtype, np.complexfloating):
        return np.dtype(np.complex128)
    raise TypeError(f"X must be a complex STFT array, got dtype {X.dtype}")


def eig_init(Cx, n_src):
    """W0 from the principal eigenvectors of the input covariance (reference overiva.py:106-109), with host LAPACK as
    in the reference.  The product paths (single GPU and bin-sharded) use the device eigensolver instead
    (``Plan.set_w_eig``: same vectors, same phase convention -- largest component real); this host form serves engines
    without one (the CPU test engine of tests/gloo_worker.py)."""
    vals, vecs = np.linalg.eig(np.asarray(Cx, dtype=np.complex128))
    F, M, _ = Cx.shape
    W0 = np.empty((F, M, n_src), dtype=np.complex128)
    for f in range(F):
        keep = np.argsort(vals[f])[-n_src:]
        W0[f] = np.conj(vecs[f][:, keep])
    return W0


def overiva(
    X,
    n_src=None,
    n_iter=20,
    proj_back=True,
    W0=None,
    model="laplace",
    init_eig=False,
    return_filters=False,
    callback=None,
):
    """
    Overdetermined IVA / AuxIVA (Scheibler & Ono 2019) on the GPU.

    Parameters
    ----------
    X: ndarray (nframes, nfrequencies, nchannels)
        STFT representation of the signal
    n_src: int, optional
        Number of sources; ``n_src == nchannels`` (default) is plain AuxIVA
    n_iter: int, optional
        Number of iterations (default 20)
    proj_back: bool, optional
        Scale by projection back onto the first channel (default True)
    W0: ndarray broadcastable to (nfrequencies, nchannels, nsrc), optional
        Initial demixing vectors (columns)
    model: str
        'laplace' (default) or 'gauss'
    init_eig: bool, optional
        Initialise from the principal eigenvectors of the input covariance when ``W0 is None``
    return_filters: bool
        Also return the demixing matrix (nfrequencies, nchannels, nsrc)
    callback: func
        Called with the current (nframes, nfrequencies, nsrc) estimate at epochs 0, 10, 20, ...

    Returns
    -------
    Y (nframes, nfrequencies, nsrc) in the dtype of X, or ``(Y, W)`` if ``return_filters``.
    """
    if not isinstance(X, DeviceX):     # (a DeviceX is a (T, F, M) complex64 tensor already on the GPU, see plan.py)
        X = np.asarray(X)
    if X.ndim != 3:
        raise ValueError("X must have shape (n_frames, n_freq, n_chan)")
    dtype = _complex_dtype(X)
    n_frames, n_freq, n_chan = X.shape
    if n_src is None:  # default to the determined case, overiva.py:83-84
        n_src = n_chan
    if not 1 <= n_src <= n_chan:
        raise ValueError(f"n_src must be in 1..{n_chan}")
    if model not in ("laplace", "gauss"):
        raise ValueError(f"model must be 'laplace' or 'gauss', got {model!r}")
    if n_iter < 0:
        raise ValueError("n_iter must be >= 0")

    args = (X, dtype, n_src, n_iter, proj_back, W0, model, init_eig, return_filters)
    called_back = []
    cb = None if callback is None else (lambda Y: (called_back.append(1), callback(Y))[1])
    try:
        return _solve(*args, cb)
    except HipError as e:
        # The plans kept between calls (up to _PLAN_CACHE_MAX, ~0.7 GB each at the headline shape) are memory the caller
        # believes free: when the device runs out, they are destroyed and the call is made once more (the library itself
        # already hands its pooled buffers back before an allocation fails, csrc/plan.hip::dev_malloc).
        if "out of memory" not in str(e).lower() or not _plan_cache or called_back:
            raise
    release_cached_buffers()
    return _solve(*args, cb)


def _solve(X, dtype, n_src, n_iter, proj_back, W0, model, init_eig, return_filters, callback):
    n_frames, n_freq, n_chan = X.shape
    precision = resolve_precision(dtype, n_chan, n_src=n_src, n_frames=n_frames)
    # (chosen by `auto` for a short frame axis, not asked for: may give way to the X-resident kernel's arithmetic, see _SingleDevice)
    short_axis_auto = _precision == "auto" and np.dtype(dtype) == np.complex64 and precision == "precise"
    group = sharded.active_group()
    if group is not None and isinstance(X, DeviceX):
        raise ValueError("a device-resident X cannot be sharded over ranks: pass the host array")
    if group is not None:
        solver = sharded.BinShardedSolver(n_frames, n_freq, n_chan, n_src, model, group=group[0], precision=precision,
                                          exchange=group[1] if len(group) > 1 else None)
    else:
        solver = _SingleDevice(n_frames, n_freq, n_chan, n_src, model, precision, prefer_resident=short_axis_auto)
    try:
        solver.set_x(X)
        solver.covariance()
        if W0 is None and init_eig:
            solver.set_w_eig()                      # overiva.py:106-109 on the device (per shard when the bins are sharded)
        else:
            solver.set_w(W0)

        epoch = 0
        while epoch < n_iter:
            if callback is not None and epoch % 10 == 0:  # overiva.py:142-148
                callback(solver.demix(proj_back, dtype))
            if callback is None:
                step = n_iter - epoch
            else:
                step = min(n_iter - epoch, 10 - epoch % 10)
            solver.iterate(step)
            epoch += step

        if group is None and n_frames * n_freq * n_src * np.dtype(dtype).itemsize >= _PREFAULT_MIN_BYTES:
            # the array the call returns: allocated, and its pages faulted in by the library's copy threads, while the GPU still
            # runs the iterations queued above (1-2 ms for the 131 MB of the headline shape, hidden behind 4 ms of kernels);
            # the hand-over then meets resident pages: 41 GB/s instead of 23-32 (tools/e2e_phases.py; csrc/host_io.hip).
            # (Beside the upload of X instead -- from a thread -- it slowed the upload by as much as it saved.)
            out = np.empty((n_frames, n_freq, n_src), dtype)
            _prefault(out)
            Y = solver.demix(proj_back, dtype, out=out)
        else:
            Y = solver.demix(proj_back, dtype)
        if return_filters:
            W = solver.get_w().astype(dtype, copy=False)
            solver.ok = True
            return Y, W
        # surface a singular solve the way numpy.linalg.solve would (overiva.py:182)
        solver.get_w()
        solver.ok = True
        return Y
    finally:
        solver.close()


_PREFAULT_MIN_BYTES = 4 << 20


def _prefault(a):
    """fault the pages of a freshly allocated array in (ctypes releases the GIL for the call)"""
    import ctypes as C

    from . import _lib

    try:
        _lib.load().oiva_host_prefault(C.c_void_p(a.ctypes.data), a.nbytes)
    except Exception:      # (an optimisation only)
        pass


# plans kept between calls (_SingleDevice.close), oldest first
_plan_cache = {}
_PLAN_CACHE_MAX = 0 if os.environ.get("OIVA_PLAN_CACHE", "1") == "0" else 2


# (device, T, F, M, K) whose X-resident launch gave up in this process (its workgroups were not all co-resident: a shared or
# CU-masked GPU): later calls of the same shape go straight to the four-launch path instead of waiting for the time-out again
_resident_gave_up = set()


class _SingleDevice:
    """all bins on one GPU"""

    # hipGraph replay pays off once an iteration is hundreds of microseconds (cheaper kernel boundaries);
    # below that eager launches are faster (measured: 513x1000x4 30.4k it/s eager vs 26.5k replayed,
    # 2048x4000x8 4.1k eager vs 4.5k replayed)
    GRAPH_MIN_ELEMENTS = 1 << 23

    def __init__(self, T, F, M, K, model, precision="fast", prefer_resident=False):
        self.key = (get_device(), T, F, M, K)
        resident_ok = os.environ.get("OIVA_RESIDENT", "1") != "0" and self.key not in _resident_gave_up
        plan = None
        if prefer_resident and precision == "precise" and M != 4 and resident_ok:
            # `auto` picked the float64 covariance for a short frame axis; where the whole iteration runs as ONE launch with X on chip
            # (2, 6, 8 channels, 1-2 sources + background) that kernel's arithmetic -- `mixed` -- stays: it has the float64 covariance
            # for 4 channels only, and the four launches in `precise` are 1.4-1.9 times its time on those shapes
            plan = Plan(T, F, M, K, model, device=get_device())
            if plan.resident_info()["qualifies"]:
                precision = "mixed"
        self.wdtype = np.complex64 if precision == "fast" else np.complex128
        self.precision = precision
        # a plan of exactly this problem kept by an earlier call (see close()): its buffers, stream and captured graphs serve again
        self.cache_key = self.key + (model, precision)
        self.ok = False
        self.plan = _plan_cache.pop(self.cache_key, None)
        if self.plan is not None:
            if plan is not None:
                plan.close()
            return
        self.plan = plan if plan is not None else Plan(T, F, M, K, model, device=get_device())
        self.plan.set_precision(precision)
        if T * F * M >= self.GRAPH_MIN_ELEMENTS:
            self.plan.use_graph(True)
        # the loop body as one persistent launch with X on chip wherever the shape qualifies (csrc/resident_kernel.inc; the
        # float64 covariance of `precise` exists there for 4 channels); $OIVA_RESIDENT=0 keeps the four-launch path
        if (precision != "precise" or M == 4) and resident_ok and self.plan.resident_info()["qualifies"]:
            self.plan.set_resident(True)

    def set_x(self, X):
        self.plan.set_x(X)

    def covariance(self):
        self.plan.covariance()

    def get_cx(self):
        return self.plan.get_cx()

    def set_w(self, W0):
        self.plan.set_w(W0)

    def set_w_eig(self):
        self.plan.set_w_eig()

    def iterate(self, n):
        self.plan.iterate(n)

    def demix(self, proj_back, dtype=np.complex64, out=None):
        return self.plan.demix(proj_back, dtype=dtype, out=out)

    def get_w(self):
        return self.plan.get_w(self.wdtype)

    def close(self):
        global _last_info
        if getattr(self.plan, "h", None):
            info = self.plan.resident_info()
            if info["fallbacks"]:
                _resident_gave_up.add(self.key)
            _last_info = {"precision": self.precision, "sharded": False, "resident_launches": info["launches"],
                          "resident_fallbacks": info["fallbacks"], "resident_give_up_code": info["last_give_up_code"]}
            # A large plan of the four-launch path is kept for the next call of the same problem instead of being destroyed:
            # creating and destroying it (streams, events, ~20 device buffers, captured graphs) is 2 ms of a 21 ms call at
            # the headline shape.  At most _PLAN_CACHE_MAX plans, only after a call that ended normally;
            # release_cached_buffers() destroys them ($OIVA_PLAN_CACHE=0: never kept).
            T, F, M = self.key[1:4]
            if (self.ok and not info["enabled"] and T * F * M >= self.GRAPH_MIN_ELEMENTS and _PLAN_CACHE_MAX > 0):
                self.plan._keep = None              # (a borrowed device X of the caller is not held on to)
                while len(_plan_cache) >= _PLAN_CACHE_MAX:
                    _plan_cache.pop(next(iter(_plan_cache))).close()
                _plan_cache[self.cache_key] = self.plan
                self.plan = None
                return
        self.plan.close()
