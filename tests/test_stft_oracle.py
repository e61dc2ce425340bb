"""The STFT oracle (oracle/stft_oracle.py) pinned independently of itself: scipy.signal.stft for framing + transform,
closed-form spectra of sinusoids, perfect reconstruction with the least-squares synthesis window, and the frozen
golden vectors.  The functions it stands for are third-party (parity unpinned, see the oracle's header).  CPU only."""
import os

import numpy as np
import pytest
import scipy.signal as ss

from conftest import GOLDEN_DIR
from oracle import stft_oracle as so


@pytest.mark.parametrize("L,hop,C,n", [(64, 32, 3, 645), (256, 64, 2, 1500), (128, 128, 1, 900), (4096, 2048, 2, 4096 * 6 + 11)])
def test_against_scipy(L, hop, C, n):
    rng = np.random.default_rng(L + hop)
    x = rng.standard_normal((n, C))
    w = so.hann(L)
    X = so.analysis(x, L, hop, w)
    T = n // hop
    assert X.shape == (T, L // 2 + 1, C)
    xp = np.concatenate([np.zeros((L - hop, C)), x])          # the zero state in front of the first sample
    for c in range(C):
        _, _, Z = ss.stft(xp[:, c], window=w, nperseg=L, noverlap=L - hop, boundary=None, padded=False)
        assert np.abs(Z.T[:T] * w.sum() - X[:, :, c]).max() < 1e-9      # scipy scales by 1 / sum(window)


def test_sinusoid_lands_in_its_bin():
    L, hop, k = 128, 64, 9
    n = L * 8
    x = np.cos(2 * np.pi * k * np.arange(n) / L)
    X = so.analysis(x, L, hop, None)
    mag = np.abs(X[3])                                        # a frame past the zero state
    assert abs(mag[k] - L / 2) < 1e-9 and np.delete(mag, k).max() < 1e-9


@pytest.mark.parametrize("L,hop", [(64, 32), (64, 16), (256, 128), (128, 128)])
def test_perfect_reconstruction(L, hop):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((L * 12 + 3, 2))
    wa = so.hann(L) if hop < L else None
    ws = so.compute_synthesis_window(wa, hop) if hop < L else None
    y = so.synthesis(so.analysis(x, L, hop, wa), L, hop, ws)
    T = x.shape[0] // hop
    assert y.shape == (T * hop, 2)
    good = T * hop - (L - hop)                                # the last frame - hop samples lack their overlap partner
    assert np.abs(y[:good] - x[:good]).max() < 1e-12


def test_golden_vectors_and_host_helpers():
    from overiva_amd import stft as st

    with np.load(os.path.join(GOLDEN_DIR, "stft_small.npz")) as d:
        for name in ("a", "b", "c"):
            x, L, hop = d[f"{name}_x"], int(d[f"{name}_L"]), int(d[f"{name}_hop"])
            wa = so.hann(L) if hop < L else None
            ws = so.compute_synthesis_window(wa, hop) if hop < L else None
            X = so.analysis(x, L, hop, wa)
            assert np.abs(X - d[f"{name}_X"]).max() < 1e-4 * np.abs(X).max()
            assert np.abs(so.synthesis(X, L, hop, ws) - d[f"{name}_y"]).max() < 1e-5
    for L, hop in ((64, 32), (96, 24), (4096, 2048)):       # the product's host-side window helpers
        assert np.allclose(st.hann(L), so.hann(L))
        assert np.allclose(st.compute_synthesis_window(st.hann(L), hop), so.compute_synthesis_window(so.hann(L), hop))
