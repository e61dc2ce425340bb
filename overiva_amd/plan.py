"""Python face of one ``oiva_plan`` (one GPU, one stream, one contiguous range of frequency bins)."""
import ctypes as C

import numpy as np

from . import _lib


def _is_f64(a):
    if a.dtype == np.complex128:
        return 1
    if a.dtype == np.complex64:
        return 0
    raise TypeError(f"complex64 or complex128 expected, got {a.dtype}")


class DeviceX:
    """A dense (T, F, M) complex64 STFT tensor that already lives in the memory of this process's GPU: ``ptr`` is the
    device address, ``owner`` whatever keeps it alive.  ``overiva()`` accepts it in place of a host array."""

    def __init__(self, ptr, shape, owner=None, host_dtype=np.complex64):
        self.ptr = int(ptr)
        self.shape = tuple(int(v) for v in shape)
        self.owner = owner
        self.dtype = np.dtype(host_dtype)      # dtype results are returned in
        self.ndim = 3


class Plan:
    """Owns the device state of one AuxIVA/OverIVA problem (or one bin shard of it).

    Mirrors the stages of reference ``overiva.py``: ``set_x`` (:132), ``covariance`` (:87),
    ``set_w`` (:89-123), ``iterate`` (:138-190), ``demix`` (:192-199), ``get_w`` (:201-202).
    """

    def __init__(self, T, F, M, K, model="laplace", device=0, F_total=None, stream=None):
        if model not in _lib.MODEL_IDS:
            # the reference silently produces NaN for an unknown model (overiva.py:152-167)
            raise ValueError(f"model must be 'laplace' or 'gauss', got {model!r}")
        self.lib = _lib.load()
        self.T, self.F, self.M, self.K = int(T), int(F), int(M), int(K)
        self.model = model
        self.F_total = int(F if F_total is None else F_total)
        self.device = int(device)
        h = C.c_void_p()
        _lib.check(self.lib.oiva_plan_create(C.byref(h), self.device, self.T, self.F, self.M, self.K,
                                             _lib.MODEL_IDS[model], self.F_total,
                                             C.c_void_p(stream) if stream else None))
        self.h = h
        self._keep = None

    # -- life cycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.oiva_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- input --------------------------------------------------------------------------------
    def set_x(self, X, f0=0):
        """X: (T, F_any, M) complex host array; uploads bins [f0, f0+F) as complex64.  A ``DeviceX`` of exactly this
        plan's shape is borrowed instead."""
        if isinstance(X, DeviceX):
            if X.shape != (self.T, self.F, self.M) or f0:
                raise ValueError(f"device X has shape {X.shape}, plan expects ({self.T}, {self.F}, {self.M})")
            return self.set_x_device(X.ptr, X.owner)
        X = np.asarray(X)
        if X.ndim != 3 or X.shape[0] != self.T or X.shape[2] != self.M or X.shape[1] < f0 + self.F:
            raise ValueError(f"X has shape {X.shape}, plan expects ({self.T}, >={f0 + self.F}, {self.M})")
        if X.dtype == np.complex128 and X.flags["C_CONTIGUOUS"]:      # converted on the device
            pitch = X.shape[1] * self.M * 16
            base = X.ctypes.data + f0 * self.M * 16
            _lib.check(self.lib.oiva_plan_set_x_host_c128(self.h, C.c_void_p(base), pitch))
            return
        if X.dtype != np.complex64 or not X.flags["C_CONTIGUOUS"]:
            X = np.ascontiguousarray(X[:, f0:f0 + self.F, :], dtype=np.complex64)
            f0 = 0
        pitch = X.shape[1] * self.M * 8
        base = X.ctypes.data + f0 * self.M * 8
        _lib.check(self.lib.oiva_plan_set_x_host(self.h, C.c_void_p(base), pitch))

    def set_x_device(self, dev_ptr, keepalive=None):
        """Borrow a dense (T, F, M) complex64 device array (e.g. a torch tensor's data_ptr())."""
        self._keep = keepalive
        _lib.check(self.lib.oiva_plan_set_x_dev(self.h, C.c_void_p(int(dev_ptr))))

    # -- prologue -----------------------------------------------------------------------------
    def covariance(self):
        _lib.check(self.lib.oiva_plan_covariance(self.h))

    def get_cx(self, dtype=np.complex128):
        """input covariance (F, M, M); the device holds it in float64"""
        out = np.empty((self.F, self.M, self.M), dtype)
        _lib.check(self.lib.oiva_plan_get_cx(self.h, _lib.ptr(out), _is_f64(out)))
        return out

    def set_w(self, W0=None):
        if W0 is None:
            _lib.check(self.lib.oiva_plan_set_w(self.h, None, 0))
            return
        W0 = np.asarray(W0)
        dt = np.complex64 if W0.dtype == np.complex64 else np.complex128
        W0 = np.ascontiguousarray(np.broadcast_to(W0, (self.F, self.M, self.K)), dtype=dt)
        _lib.check(self.lib.oiva_plan_set_w(self.h, _lib.ptr(W0), _is_f64(W0)))

    def set_w_pca(self, return_eigenvalues=False):
        """W := eigenvectors of the K largest eigenvalues of the input covariance, ascending (``eigh``'s
        ``w[:, :, -K:]``, auxiva_pca.py:75-81), from the Jacobi eigensolver on the device; optionally returns all
        eigenvalues (F, M) ascending."""
        ev = np.empty((self.F, self.M), np.float64) if return_eigenvalues else None
        _lib.check(self.lib.oiva_plan_set_w_pca(self.h, _lib.ptr(ev) if ev is not None else None))
        return ev

    def set_w_eig(self):
        """``init_eig`` of overiva.py:106-109 on the device: W := conj of the K principal eigenvectors of the input
        covariance with LAPACK's phase convention (largest component real)"""
        _lib.check(self.lib.oiva_plan_set_w_eig(self.h))

    # -- iteration ----------------------------------------------------------------------------
    def iterate(self, n=1):
        _lib.check(self.lib.oiva_plan_iterate(self.h, int(n)))

    def power(self):
        _lib.check(self.lib.oiva_plan_power(self.h))

    BINS_PER_PART = 64   # one partial-power part per batch of 64 bins (power kernel workgroup)

    @classmethod
    def power_parts(cls, n_bins):
        return (n_bins + cls.BINS_PER_PART - 1) // cls.BINS_PER_PART

    def power_buffer(self, parts_per_rank):
        p, nbytes = C.c_void_p(), C.c_longlong()
        _lib.check(self.lib.oiva_plan_power_buffer(self.h, int(parts_per_rank), C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    def update(self, parts_dev_ptr, nparts):
        _lib.check(self.lib.oiva_plan_update(self.h, C.c_void_p(int(parts_dev_ptr)), int(nparts)))

    # -- OGIVE (reference ive.py) ------------------------------------------------------------------
    def ogive_begin(self, update="demix", model="laplace"):
        _lib.check(self.lib.oiva_plan_ogive_begin(self.h, {"demix": 0, "mix": 1, "switching": 2}[update],
                                                  _lib.MODEL_IDS[model]))

    def ogive_iterate(self, first_epoch, n, step_size=0.1, tol=1e-3):
        """up to n epochs; returns (epochs that changed the state, stopping rule met, max ||delta|| of the last epoch)"""
        ran, conv, md = C.c_int(), C.c_int(), C.c_double()
        _lib.check(self.lib.oiva_plan_ogive_iterate(self.h, int(first_epoch), int(n), float(step_size), float(tol),
                                                    C.byref(ran), C.byref(conv), C.byref(md)))
        return ran.value, bool(conv.value), md.value

    def iterate_timed(self, n, per_kernel=False):
        total = C.c_float()
        if per_kernel:
            arr = (C.c_float * _lib.N_STAGES)()
            _lib.check(self.lib.oiva_plan_iterate_timed(self.h, int(n), C.byref(total), arr))
            return total.value, dict(zip(_lib.STAGE_NAMES, list(arr)))
        _lib.check(self.lib.oiva_plan_iterate_timed(self.h, int(n), C.byref(total), None))
        return total.value, None

    # -- epilogue -----------------------------------------------------------------------------
    def demix(self, proj_back=True, out=None, f0=0, dtype=np.complex64):
        """Y (T, F, K) complex64 (or complex128: converted on the device); with ``out`` (T, F_any, K) writes bins
        [f0, f0+F) in place, in the dtype of ``out``."""
        if out is None:
            out = np.empty((self.T, self.F, self.K), dtype)
            f0 = 0
        assert out.dtype in (np.complex64, np.complex128) and out.flags["C_CONTIGUOUS"]
        size = out.dtype.itemsize
        pitch = out.shape[1] * self.K * size
        base = out.ctypes.data + f0 * self.K * size
        fn = self.lib.oiva_plan_demix if size == 8 else self.lib.oiva_plan_demix_c128
        _lib.check(fn(self.h, C.c_void_p(base), pitch, 1 if proj_back else 0))
        return out

    def set_io_slab(self, nbytes=0):
        """bytes of output per slab of the pipelined hand-over of ``demix`` (0: the default); test hook"""
        _lib.check(self.lib.oiva_plan_set_io_slab(self.h, int(nbytes)))

    def demix_device(self, proj_back=False):
        """Y stays on the device: a ``DeviceX`` (T, F, K) that another plan can take as its input (``set_x``); valid
        until this plan's next demix or its close()."""
        dev = C.c_void_p()
        _lib.check(self.lib.oiva_plan_demix_dev(self.h, 1 if proj_back else 0, C.byref(dev)))
        return DeviceX(dev.value, (self.T, self.F, self.K), owner=self)

    def get_w(self, dtype=np.complex64):
        out = np.empty((self.F, self.M, self.K), dtype)
        _lib.check(self.lib.oiva_plan_get_w(self.h, _lib.ptr(out), _is_f64(out)))
        return out

    def sync(self):
        _lib.check(self.lib.oiva_plan_sync(self.h))

    def save_w(self):
        """keep a copy of the demixing state on the device (asynchronous, ordered on the plan's stream)"""
        _lib.check(self.lib.oiva_plan_save_w(self.h))

    def restore_w(self):
        """bring back the state kept by ``save_w``"""
        _lib.check(self.lib.oiva_plan_restore_w(self.h))

    # -- knobs --------------------------------------------------------------------------------
    def cov_splits(self):
        n = C.c_int()
        _lib.check(self.lib.oiva_plan_get_cov_splits(self.h, C.byref(n)))
        return n.value

    def set_cov_splits(self, n):
        _lib.check(self.lib.oiva_plan_set_cov_splits(self.h, int(n)))

    def set_cov_quad(self, enable=True):
        """covariance pass of a 9..16-channel plan: the vector-ALU kernels (default; odd counts on a zero-padded copy of X; float32 products: four lanes per
        (bin, frame) for <= 4 sources, 32 lanes and all sources in one pass for more; ``precise``: the float64 form of the
        latter for >= 3 sources) or the planar matrix-core kernel; returns whether a vector-ALU kernel is now active"""
        a = C.c_int()
        _lib.check(self.lib.oiva_plan_set_cov_quad(self.h, 1 if enable else 0, C.byref(a)))
        return bool(a.value)

    def set_cov_hmfma(self, enable=True):
        """9..16 sources on 10..16 channels: the sources on the fp32 matrix cores (default) or the vector-ALU kernel alone"""
        _lib.check(self.lib.oiva_plan_set_cov_hmfma(self.h, 1 if enable else 0))

    def set_fuse_cov_update(self, enable=True):
        """covariance + per-bin update as one launch where the shape qualifies (8 channels, 2 sources, four frame splits: the
        headline shape); returns whether this plan's iterations run it.  ``enable=None`` only asks."""
        a = C.c_int()
        _lib.check(self.lib.oiva_plan_set_fuse_cov_update(self.h, -1 if enable is None else (1 if enable else 0), C.byref(a)))
        return bool(a.value)

    def set_pow_splits(self, n):
        _lib.check(self.lib.oiva_plan_set_pow_splits(self.h, int(n)))

    def use_graph(self, enable=True):
        _lib.check(self.lib.oiva_plan_use_graph(self.h, 1 if enable else 0))

    def set_precision(self, mode="fast", row_layout=False):
        """``"fast"`` (float32 products, lane chains and per-bin algebra), ``"mixed"`` (float32 products and lane
        chains of the covariance pass, float64 sums across lanes / splits and float64 per-bin algebra, W_hat carried in
        complex128), ``"precise"`` (float64 sums of exact float64 products in the covariance pass + float64 algebra: the
        reference's complex128 arithmetic on complex64 data), or an int of ``_lib.PREC_*`` bits."""
        flags = _lib.PREC_BY_NAME[mode] if isinstance(mode, str) else int(mode)
        if row_layout:
            flags |= _lib.PREC_UPDATE_ROWS
        _lib.check(self.lib.oiva_plan_set_precision(self.h, flags))

    # -- X-resident iteration (include/overiva_hip.h, csrc/resident_kernel.inc) -------------------
    RESIDENT_INFO = ("qualifies", "enabled", "bin_groups", "frame_splits", "frames_per_split", "frames_per_lane",
                     "frames_in_registers", "lds_bytes", "last_give_up_code", "launches", "fallbacks", "x_bytes_per_cu")
    RESIDENT_PHASES = ("demix_power", "parts", "activation", "cov_accumulate", "cov_reduce", "wait_partials", "ip_update", "wait_w")

    def set_resident(self, enable=True):
        """the loop body as ONE persistent launch per ``iterate`` call with X held on chip; ValueError when the shape
        does not qualify"""
        _lib.check(self.lib.oiva_plan_set_resident(self.h, 1 if enable else 0))

    def resident_info(self):
        arr = (C.c_int * len(self.RESIDENT_INFO))()
        _lib.check(self.lib.oiva_plan_resident_info(self.h, arr))
        return dict(zip(self.RESIDENT_INFO, list(arr)))

    def resident_phases(self):
        """(microseconds per phase of workgroup 0 averaged over the last resident launch, iterations covered)"""
        arr = (C.c_double * len(self.RESIDENT_PHASES))()
        n = C.c_int()
        _lib.check(self.lib.oiva_plan_resident_phases(self.h, arr, C.byref(n)))
        return dict(zip(self.RESIDENT_PHASES, list(arr))), n.value

    def set_resident_splits(self, nsplit=0):
        _lib.check(self.lib.oiva_plan_set_resident_splits(self.h, int(nsplit)))

    def resident_connect(self, xchg_handle):
        """bins sharded over GPUs: the connected exchange (``exchange.PushExchange.h``) the resident kernel pushes its
        rank's partial powers through; None disconnects"""
        _lib.check(self.lib.oiva_plan_resident_connect(self.h, xchg_handle))

    def fused_connect(self, xchg_handle):
        """bins sharded over GPUs and the shard does not fit on chip: the connected exchange (``exchange.PushExchange.h``, slot
        = T * K * 8 bytes) through which the ACTIVATION kernel of the four-launch iteration exchanges the ranks' partial powers
        itself; ``iterate`` then works on a shard, replaying captured graphs.  None disconnects."""
        _lib.check(self.lib.oiva_plan_fused_connect(self.h, xchg_handle))

    def fused_loopback(self, world=2):
        """one GPU plays all ``world`` ranks of that exchange against itself (the other ranks' sums are zeros); 0 switches it off"""
        _lib.check(self.lib.oiva_plan_fused_loopback(self.h, int(world)))

    def fused_debug(self, timeout_ms=0, stall=False):
        _lib.check(self.lib.oiva_plan_fused_debug(self.h, int(timeout_ms), 1 if stall else 0))

    def resident_trace(self, enable=True, fetch=False):
        """diagnostics: record every workgroup's timestamps in the next launches (<= 64 iterations); with ``fetch`` returns the
        last launch's (n_wg, n_iter, 16) array of 100 MHz ticks"""
        nw, ni = C.c_int(), C.c_int()
        _lib.check(self.lib.oiva_plan_resident_trace(self.h, 1 if enable else 0, None, C.byref(nw), C.byref(ni)))
        if not fetch or ni.value == 0:
            return None
        out = np.zeros((nw.value, ni.value, 16), np.uint64)
        _lib.check(self.lib.oiva_plan_resident_trace(self.h, 1 if enable else 0, _lib.ptr(out), C.byref(nw), C.byref(ni)))
        return out

    def resident_debug(self, timeout_ms=0, stall_block=-1, from_iteration=0):
        """test hooks: time-out of a wait, and a workgroup that stops publishing from an iteration of the launch on"""
        _lib.check(self.lib.oiva_plan_resident_debug_from(self.h, int(timeout_ms), int(stall_block), int(from_iteration)))

    def resident_loopback(self, world=8):
        """one GPU plays all ``world`` ranks of the kernel's multi-GPU exchange against itself (leader gather, ``world`` slot
        stores, rank-order sum; the other ranks' sums are exact zeros, so the result is that of one rank); 0 switches it off.
        Call it while resident is off."""
        _lib.check(self.lib.oiva_plan_resident_loopback(self.h, int(world)))

    # -- test-only stage access -----------------------------------------------------------------
    def t_set_rinv(self, rinv):
        rinv = np.ascontiguousarray(rinv, dtype=np.float32)
        assert rinv.shape == (self.T, self.K)
        _lib.check(self.lib.oiva_test_set_rinv(self.h, _lib.ptr(rinv)))

    def t_get_rinv(self):
        r = np.empty((self.T, self.K), np.float32)
        w = np.empty((self.K,), np.float32)
        _lib.check(self.lib.oiva_test_get_rinv(self.h, _lib.ptr(r), _lib.ptr(w)))
        return r, w

    def t_run_weighted_cov(self):
        _lib.check(self.lib.oiva_test_run_weighted_cov(self.h))

    def t_get_v(self, dtype=np.complex64):
        v = np.empty((self.K, self.F, self.M, self.M), dtype)
        _lib.check(self.lib.oiva_test_get_v(self.h, _lib.ptr(v), _is_f64(v)))
        return v

    def t_run_update(self):
        _lib.check(self.lib.oiva_test_run_update(self.h))

    def t_get_what(self, dtype=np.complex64):
        w = np.empty((self.F, self.M, self.M), dtype)
        _lib.check(self.lib.oiva_test_get_what(self.h, _lib.ptr(w), _is_f64(w)))
        return w

    def t_set_what(self, What):
        What = np.asarray(What)
        What = np.ascontiguousarray(What, dtype=np.complex64 if What.dtype == np.complex64 else np.complex128)
        assert What.shape == (self.F, self.M, self.M)
        _lib.check(self.lib.oiva_test_set_what(self.h, _lib.ptr(What), _is_f64(What)))

    def t_time_stage(self, stage, reps=20):
        ms = C.c_float()
        _lib.check(self.lib.oiva_test_time_stage(self.h, _lib.STAGE_NAMES.index(stage), int(reps), C.byref(ms)))
        return ms.value

    def t_run_power(self):
        p = np.empty((self.T, self.K), np.float32)
        _lib.check(self.lib.oiva_test_run_power(self.h, _lib.ptr(p)))
        return p
