"""
ORACLE -- test infrastructure only.  NOT part of the product path.

NumPy (float64) restatement of the STFT analysis / synthesis the reference's drivers call:

    X = pra.transform.analysis(mics_signals.T, framesize, framesize // 2, win=win_a)     overiva_oneshot.py:293-295
    y = pra.transform.synthesis(Y, framesize, framesize // 2, win=win_s)                 overiva_oneshot.py:371-379
    win_a = pra.hann(framesize); win_s = pra.transform.compute_synthesis_window(win_a, framesize // 2)   :157-158

PARITY UNPINNED: these are functions of the third-party dependency ``pyroomacoustics==0.1.23``
(reference ``environment.yml:14``) whose source is absent from ``/root/reference`` and from this image, and the
reference holds no test or golden vector for them.  What is restated is that library's published block-processing
convention: every frame holds ``hop`` new samples behind ``frame - hop`` samples of the previous frames, the state
before the first sample is zero, hence ``n_frames = n_samples // hop``; synthesis overlap-adds the windowed inverse
transforms and returns ``n_frames * hop`` samples.  Independent pins used by tests/test_stft_oracle.py instead:
``scipy.signal.stft`` on the zero-prefixed signal (framing + transform), closed-form transforms of sinusoids, and
perfect reconstruction with the least-squares synthesis window.
"""
import numpy as np


def hann(n):
    """periodic Hann window, 0.5 * (1 - cos(2 pi k / n))"""
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(n) / n))


def compute_synthesis_window(win_a, hop):
    """least-squares optimal synthesis window for analysis window ``win_a`` at hop ``hop``:
    win_a / (sum over all shifts by multiples of hop of win_a^2)"""
    win_a = np.asarray(win_a, dtype=np.float64)
    L = win_a.shape[0]
    norm = np.zeros(L)
    n = 0
    while n - hop > -L:
        n -= hop
    while n < L:
        if n == 0:
            norm += win_a ** 2
        elif n < 0:
            norm[: n + L] += win_a[-n - L:] ** 2
        else:
            norm[n:] += win_a[:-n] ** 2
        n += hop
    return win_a / norm


def analysis(x, L, hop, win=None):
    """x (n_samples, n_chan) -> X (n_frames, L // 2 + 1, n_chan) complex128"""
    x = np.asarray(x, dtype=np.float64)
    mono = x.ndim == 1
    if mono:
        x = x[:, None]
    T = x.shape[0] // hop
    xp = np.concatenate([np.zeros((L - hop, x.shape[1])), x], axis=0)
    X = np.empty((T, L // 2 + 1, x.shape[1]), dtype=np.complex128)
    w = np.ones(L) if win is None else np.asarray(win, dtype=np.float64)
    for t in range(T):
        X[t] = np.fft.rfft(xp[t * hop: t * hop + L] * w[:, None], axis=0)
    return X[:, :, 0] if mono else X


def synthesis(X, L, hop, win=None):
    """X (n_frames, L // 2 + 1, n_chan) -> x (n_frames * hop, n_chan) float64"""
    X = np.asarray(X)
    mono = X.ndim == 2
    if mono:
        X = X[:, :, None]
    T, _, C = X.shape
    w = np.ones(L) if win is None else np.asarray(win, dtype=np.float64)
    out = np.zeros((L - hop + T * hop + L, C))
    for t in range(T):
        out[t * hop: t * hop + L] += np.fft.irfft(X[t], n=L, axis=0) * w[:, None]
    y = out[L - hop: L - hop + T * hop]
    return y[:, 0] if mono else y
