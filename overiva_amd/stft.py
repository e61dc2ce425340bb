"""STFT analysis / synthesis on the GPU (hipFFT) with the call shapes of the functions the reference's drivers use:

    X = pra.transform.analysis(mics_signals.T, framesize, framesize // 2, win=win_a)     overiva_oneshot.py:293-295
    y = pra.transform.synthesis(Y, framesize, framesize // 2, win=win_s)                 overiva_oneshot.py:371-379

so that ``from overiva_amd import stft as transform`` is a drop-in for those two calls (and ``hann`` /
``compute_synthesis_window`` for ``overiva_oneshot.py:157-158``).  The third-party originals are absent from the
reference tree (parity unpinned, see oracle/stft_oracle.py); the convention restated: ``hop`` new samples per frame
behind ``frame - hop`` old ones, zero state, ``n_frames = n_samples // hop``.  Arithmetic is float32 / complex64 on the
device; the result is cast to the dtype the reference's call produces (complex128 / float64 for float64 input).
"""
import ctypes as C

import numpy as np

from . import _lib


def hann(n):
    """periodic Hann window (host)"""
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n) / n))


def compute_synthesis_window(analysis_window, hop):
    """least-squares optimal synthesis window: analysis window over the sum of its squared shifts by multiples of hop"""
    w = np.asarray(analysis_window, dtype=np.float64)
    L = w.shape[0]
    norm = np.zeros(L)
    n = -((L - 1) // hop) * hop
    while n < L:
        if n == 0:
            norm += w ** 2
        elif n < 0:
            norm[: n + L] += w[-n - L:] ** 2
        else:
            norm[n:] += w[:-n] ** 2
        n += hop
    return w / norm


class STFT:
    """one handle = one (n_samples, n_chan, frame, hop, windows) configuration on one GPU"""

    def __init__(self, n_samples, n_chan, L, hop, win_a=None, win_s=None, device=0):
        self.lib = _lib.load()
        self.n_samples, self.n_chan, self.L, self.hop = int(n_samples), int(n_chan), int(L), int(hop)
        wa = None if win_a is None else np.ascontiguousarray(win_a, dtype=np.float32)
        ws = None if win_s is None else np.ascontiguousarray(win_s, dtype=np.float32)
        for w in (wa, ws):
            if w is not None and w.shape != (self.L,):
                raise ValueError("window length must equal the frame length")
        h = C.c_void_p()
        _lib.check(self.lib.oiva_stft_create(C.byref(h), int(device), self.n_samples, self.n_chan, self.L, self.hop,
                                             None if wa is None else _lib.ptr(wa), None if ws is None else _lib.ptr(ws)))
        self.h = h
        t, f = C.c_int(), C.c_int()
        _lib.check(self.lib.oiva_stft_shape(self.h, C.byref(t), C.byref(f)))
        self.n_frames, self.n_freq = t.value, f.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.oiva_stft_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def analysis(self, x, to_host=True):
        """x (n_samples, n_chan) real -> X (n_frames, n_freq, n_chan) complex64; with ``to_host=False`` returns the
        device pointer of X instead (for ``Plan.set_x_device``), valid until the next call on this handle"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.shape != (self.n_samples, self.n_chan):
            raise ValueError(f"x has shape {x.shape}, expected {(self.n_samples, self.n_chan)}")
        dev = C.c_void_p()
        if to_host:
            X = np.empty((self.n_frames, self.n_freq, self.n_chan), np.complex64)
            _lib.check(self.lib.oiva_stft_analysis(self.h, _lib.ptr(x), _lib.ptr(X), C.byref(dev)))
            return X
        _lib.check(self.lib.oiva_stft_analysis(self.h, _lib.ptr(x), None, C.byref(dev)))
        return dev.value

    def analysis_device(self, x):
        """as ``analysis``, but X stays in device memory: a ``DeviceX`` that ``overiva()`` / ``auxiva_pca()`` /
        ``Plan.set_x`` take in place of a host array (valid until the next call on this handle)"""
        from .plan import DeviceX

        return DeviceX(self.analysis(x, to_host=False), (self.n_frames, self.n_freq, self.n_chan), owner=self)

    def synthesis(self, Y):
        """Y (n_frames, n_freq, k) complex, k <= n_chan -> y (n_frames * hop, k) float32"""
        Y = np.ascontiguousarray(Y, dtype=np.complex64)
        if Y.ndim != 3 or Y.shape[:2] != (self.n_frames, self.n_freq) or Y.shape[2] > self.n_chan:
            raise ValueError(f"Y has shape {Y.shape}, expected ({self.n_frames}, {self.n_freq}, <= {self.n_chan})")
        y = np.empty((self.n_frames * self.hop, Y.shape[2]), np.float32)
        _lib.check(self.lib.oiva_stft_synthesis(self.h, _lib.ptr(Y), Y.shape[2], _lib.ptr(y)))
        return y


def _device():
    from .overiva import get_device

    return get_device()


def analysis(x, L, hop, win=None, zp_back=0, zp_front=0):
    """``pra.transform.analysis``: x (n_samples,) or (n_samples, n_chan) -> (n_frames, L // 2 + 1[, n_chan])"""
    if zp_back or zp_front:
        raise NotImplementedError("zero padding of the frames is not supported")
    x = np.asarray(x)
    mono = x.ndim == 1
    x2 = x[:, None] if mono else x
    out_dtype = np.complex128 if x.dtype == np.float64 else np.complex64
    with STFT(x2.shape[0], x2.shape[1], L, hop, win_a=win, device=_device()) as s:
        X = s.analysis(x2).astype(out_dtype, copy=False)
    return X[:, :, 0] if mono else X


def synthesis(X, L, hop, win=None, zp_back=0, zp_front=0):
    """``pra.transform.synthesis``: X (n_frames, L // 2 + 1[, n_chan]) -> (n_frames * hop[, n_chan])"""
    if zp_back or zp_front:
        raise NotImplementedError("zero padding of the frames is not supported")
    X = np.asarray(X)
    mono = X.ndim == 2
    X3 = X[:, :, None] if mono else X
    out_dtype = np.float64 if X.dtype == np.complex128 else np.float32
    T = X3.shape[0]
    with STFT(T * hop, X3.shape[2], L, hop, win_s=win, device=_device()) as s:
        y = s.synthesis(X3).astype(out_dtype, copy=False)
    return y[:, 0] if mono else y
