"""The two reformulations of the per-bin update the float64 kernels use (csrc/kernels_update.hip: update_det_kernel,
update_gram_kernel; csrc/kernels_update16.hip: update_det16_kernel), restated in NumPy and held against the oracle's
restatement of the reference's own chain (overiva.py:176-190: a solve with W_hat^H V_s per source, J after every source).
CPU only: it pins the ALGEBRA the kernels rely on; the kernels themselves are compared with the oracle in the -m gpu tests.
"""
import numpy as np
import pytest

from oracle import overiva_oracle as orc


def _case(F, M, K, seed):
    rng = np.random.default_rng(seed)
    T = 4 * M + 7
    X = (rng.normal(size=(T, F, M)) + 1j * rng.normal(size=(T, F, M))) @ (rng.normal(size=(M, M)) + 1j * rng.normal(size=(M, M)))
    Cx = orc.input_covariance(X)
    W_hat = orc.init_demixing(Cx, K, W0=(rng.normal(size=(F, M, K)) + 1j * rng.normal(size=(F, M, K))))
    rinv = rng.gamma(2.0, 1.0, (T, K))
    V = orc.weighted_cov_all(X, rinv)
    return W_hat, V, Cx


def _herm(a):
    return np.conj(np.swapaxes(a, -1, -2))


def determined_by_maintained_inverse(W_hat, V):
    """K = M:  w = V_s^-1 u, u = column s of C = (W_hat^H)^-1;  d = u^H w (= w^H V_s w);  C follows the new row s of W_hat^H by
    the rank-one formula  C' = C - (u / d) (w^H C - sqrt(d) e_s^T)  with the w BEFORE normalisation."""
    W_hat = W_hat.copy()
    F, M, _ = W_hat.shape
    C = np.linalg.inv(_herm(W_hat))
    for s in range(M):
        u = C[:, :, s]
        w = np.linalg.solve(V[s], u[..., None])[..., 0]
        y = np.einsum("fi,fij->fj", np.conj(w), C)
        d = y[:, s].real.copy()
        assert np.allclose(d, np.einsum("fi,fi->f", np.conj(u), w).real)
        W_hat[:, :, s] = w / np.sqrt(d)[:, None]
        y[:, s] -= np.sqrt(d)
        C = C - (u / d[:, None])[:, :, None] * y[:, None, :]
        # the maintained matrix IS the inverse of the updated W_hat^H
        assert np.allclose(C @ _herm(W_hat), np.eye(M)[None], atol=1e-9)
    return W_hat


def overdetermined_by_gram_form(W_hat, V, Cx, K):
    """K < M:  column s of (W_hat^H)^-1 = P G^-1 e_s with P = Cx W, G = W^H Cx W;  w = V_s^-1 c, w /= sqrt(c^H w);  J once, from the
    final W (overiva.py:96-98)."""
    W_hat = W_hat.copy()
    W = W_hat[:, :, :K].copy()
    for s in range(K):
        P = Cx @ W
        G = _herm(W) @ P
        c = P @ np.linalg.inv(G)[:, :, s][..., None]
        w = np.linalg.solve(V[s], c)[..., 0]
        d = np.einsum("fi,fi->f", np.conj(c[..., 0]), w).real
        W[:, :, s] = w / np.sqrt(d)[:, None]
    W_hat[:, :, :K] = W
    W_hat[:, :K, K:] = orc.orth_constraint_J(W, Cx, K)
    return W_hat


@pytest.mark.parametrize("M", [2, 3, 5, 8, 12, 16])
def test_determined_update_through_the_maintained_inverse(M):
    W_hat, V, Cx = _case(3, M, M, seed=M)
    ref = orc.ip_update_bin(W_hat, V, Cx, M)
    got = determined_by_maintained_inverse(W_hat, V)
    assert orc.rel_err(got, ref) < 1e-10


@pytest.mark.parametrize("shape", [(4, 3), (5, 3), (6, 4), (8, 3), (8, 4), (8, 6), (7, 5), (3, 1), (8, 2), (16, 4)])
def test_update_with_background_through_the_gram_form(shape):
    M, K = shape
    W_hat, V, Cx = _case(3, M, K, seed=10 * M + K)
    ref = orc.ip_update_bin(W_hat, V, Cx, K)
    got = overdetermined_by_gram_form(W_hat, V, Cx, K)
    assert orc.rel_err(got, ref) < 1e-10
    # the identity itself, on the state the chain starts from
    C = np.linalg.inv(_herm(W_hat))
    W = W_hat[:, :, :K]
    G = _herm(W) @ Cx @ W
    assert np.allclose(C[:, :, :K], Cx @ W @ np.linalg.inv(G), atol=1e-9)
