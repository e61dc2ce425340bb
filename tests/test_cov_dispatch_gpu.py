"""Every (channels, sources, arithmetic) combination reaches SOME covariance kernel (csrc/kernels_cov*.hip: one lane per
(bin, frame) with or without the LDS-DMA ring, 2 / 4 / 32 lanes per (bin, frame), float64 forms, the matrix-core kernels): the
weighted covariance of all sources (reference overiva.py:179) and the input covariance (:87) against the oracle on one small
ragged shape per combination, so that no dispatch rule can route a shape to a kernel that does not handle it."""
import numpy as np
import pytest

from oracle import overiva_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oa():
    import overiva_amd
    from overiva_amd import _lib

    _lib.load()
    return overiva_amd


def _cases():
    out = []
    for M in range(1, 17):
        for K in sorted({k for k in (1, 2, 3, 4, 5, 8, 12, M) if k <= M}):
            out.append((M, K))
    return out


@pytest.mark.parametrize("mode", ["fast", "mixed", "precise"])
@pytest.mark.parametrize("case", _cases(), ids=lambda c: f"{c[0]}ch{c[1]}src")
def test_every_shape_has_a_covariance_kernel(oa, case, mode):
    M, K = case
    T, F = 71 + 3 * M, 35 - M          # frames that are no multiple of any step, bins that are no multiple of 2 / 16 / 32
    X = orc.synth_mixture(T, F, M, K, seed=100 * M + K)
    rinv = np.random.default_rng(M + 17 * K).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision(mode)
        p.set_x(X)
        p.covariance()
        Cx = p.get_cx()
        p.t_set_rinv(rinv)
        p.t_run_weighted_cov()
        V = p.t_get_v(np.complex128)
    w = 1.0 / (np.float32(1) / rinv).astype(np.float64) if mode == "precise" else rinv.astype(np.float64)
    eV = orc.rel_err(V, orc.weighted_cov_all(X, w))
    eC = orc.rel_err(Cx, orc.input_covariance(X.astype(np.complex128)))
    # float64 accumulation: exact products (9..16 channels: the weights travel through a float32 table); float32: chains
    tol = (1e-12 if M <= 8 else 2e-7) if mode == "precise" else 3e-7
    assert eV < tol and eC < tol, (eV, eC)
    assert np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))
