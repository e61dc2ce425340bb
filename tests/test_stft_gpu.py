"""STFT analysis / synthesis on the GPU (hipFFT, overiva_amd/stft.py -> oiva_stft_*) against the NumPy oracle and the
frozen golden vectors, plus the time-domain-in / time-domain-out chain around the solver.  Replaces the reference
drivers' pra.transform.analysis / synthesis calls (overiva_oneshot.py:293-295,371-379; third-party, parity unpinned).
Needs an MI355X: run with ``-m gpu``."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import overiva_oracle as orc
from oracle import stft_oracle as so

pytestmark = pytest.mark.gpu
TOL = 1e-5      # float32 FFT of length <= 4096 against float64


@pytest.mark.parametrize("L,hop,C,n", [(64, 32, 3, 645), (256, 64, 2, 1500), (128, 128, 1, 900), (4096, 2048, 5, 4096 * 20 + 777),
                                       (512, 256, 8, 512 * 40), (4096, 2048, 1, 2048)])
def test_analysis_and_synthesis_match_the_oracle(L, hop, C, n):
    from overiva_amd import stft as st

    rng = np.random.default_rng(L + C)
    x = rng.standard_normal((n, C))
    wa = so.hann(L) if hop < L else None
    ws = so.compute_synthesis_window(wa, hop) if hop < L else None
    X = st.analysis(x, L, hop, win=wa)
    Xr = so.analysis(x, L, hop, wa)
    assert X.shape == Xr.shape and X.dtype == np.complex128          # float64 in -> complex128 out, like the reference's call
    assert orc.rel_err(X, Xr) < TOL
    y = st.synthesis(Xr, L, hop, win=ws)
    yr = so.synthesis(Xr, L, hop, ws)
    assert y.shape == yr.shape and y.dtype == np.float64
    assert orc.rel_err(y, yr) < TOL
    # float32 in -> complex64 out; mono input keeps its 2-D shape
    X32 = st.analysis(x[:, 0].astype(np.float32), L, hop, win=wa)
    assert X32.dtype == np.complex64 and X32.shape == Xr.shape[:2] and orc.rel_err(X32, Xr[:, :, 0]) < TOL


def test_golden_vectors():
    from overiva_amd import stft as st

    with np.load(os.path.join(GOLDEN_DIR, "stft_small.npz")) as d:
        for name in ("a", "b", "c"):
            x, L, hop = d[f"{name}_x"], int(d[f"{name}_L"]), int(d[f"{name}_hop"])
            wa = so.hann(L) if hop < L else None
            ws = so.compute_synthesis_window(wa, hop) if hop < L else None
            assert orc.rel_err(st.analysis(x, L, hop, win=wa), d[f"{name}_X"]) < TOL
            assert orc.rel_err(st.synthesis(d[f"{name}_X"], L, hop, win=ws), d[f"{name}_y"]) < TOL


def test_errors():
    from overiva_amd import stft as st

    with pytest.raises(ValueError):
        st.analysis(np.zeros((100, 2)), 63, 32)            # odd frame
    with pytest.raises(ValueError):
        st.analysis(np.zeros((10, 2)), 64, 32)             # fewer samples than one hop
    with pytest.raises(NotImplementedError):
        st.analysis(np.zeros((100, 2)), 64, 32, zp_back=8)


def test_audio_in_audio_out_separation():
    """time-domain mixture -> STFT (GPU) -> overiva (GPU, X handed over on the device) -> iSTFT (GPU): the chain
    of overiva_oneshot.py:293-379 with no host-side transform; the separated signals must beat the mixture's SIR
    in the time domain and agree with the host-array path"""
    import overiva_amd as oa
    from overiva_amd import stft as st

    L, hop, M, K, n = 512, 256, 4, 2, 256 * 400
    rng = np.random.default_rng(4)
    env = np.repeat(rng.gamma(0.3, 1.0, (n // 512, K)), 512, axis=0)[:n]          # slowly varying source activity
    src = env * rng.standard_normal((n, K))
    A = rng.standard_normal((M, K))
    A[:K] += 2 * np.eye(K)
    x = src @ A.T + 0.01 * rng.standard_normal((n, M))
    wa = st.hann(L)
    ws = st.compute_synthesis_window(wa, hop)
    with st.STFT(n, M, L, hop, win_a=wa, win_s=ws) as s:
        T, F = s.n_frames, s.n_freq
        X_dev = s.analysis(x, to_host=False)                                       # stays in HBM
        with oa.Plan(T, F, M, K, "laplace") as p:
            p.set_precision("precise")
            p.set_x_device(X_dev)
            p.covariance()
            p.set_w(None)
            p.iterate(30)
            Y = p.demix(proj_back=True)
        y = s.synthesis(Y)
        X_host = s.analysis(x)
        Y3 = oa.overiva(s.analysis_device(x), n_src=K, n_iter=30, proj_back=True)  # the drop-in call on a device tensor
    Y2 = oa.overiva(X_host, n_src=K, n_iter=30, proj_back=True)
    assert orc.rel_err(Y, Y2) < 1e-5                                               # device hand-over == host path
    assert np.array_equal(Y3, Y2)
    assert y.shape == (T * hop, K)

    def sir(sig):                                                                  # best-permutation SIR via projections on the sources
        G = np.linalg.lstsq(src[: len(sig)], sig, rcond=None)[0]                  # (K_src, K_out)
        P = (G ** 2) * np.sum(src[: len(sig)] ** 2, axis=0)[:, None]
        return max(np.mean([10 * np.log10(P[perm[k], k] / (P[:, k].sum() - P[perm[k], k])) for k in range(K)])
                   for perm in ((0, 1), (1, 0)))

    sir_in, sir_out = sir(x[: T * hop, :K]), sir(y)
    print(f"\n[stft] time-domain SIR {sir_in:.1f} dB -> {sir_out:.1f} dB")
    assert sir_out > sir_in + 10.0
