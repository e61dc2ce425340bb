"""
ORACLE -- test infrastructure only.  NOT part of the product path.

NumPy restatement of ``ogive()`` (reference ``ive.py:33-256``): orthogonally constrained independent vector
extraction of ONE source by gradient steps (Koldovsky & Tichavsky 2018), the fourth algorithm of the reference's
sweep (``overiva_sim.py:313-315``, ``overiva_sim_config.json:77-100``).

Pinning status: PINNED on outputs of the real ``/root/reference/ive.py`` recorded by
``tests/golden/make_ogive_golden.py`` (the reference file is imported unmodified; besides the ``pyroomacoustics``
stub of overiva_oracle.py it needs ``np.bool``, removed from NumPy >= 1.24, which the generator supplies through a
module-global proxy).  ``projection_back`` stays parity-unpinned as in overiva_oracle.py.

``ogive_faithful`` keeps the reference's statement order and dtypes; ``ogive_staged`` cuts the iteration where the
GPU kernels do: power pass -> activation -> weighted covariance V (``x_psi = V w / (w^H V w)``, because
``sum_t r_inv x conj(y) = T V w`` and ``zeta = sum_t r_inv |y|^2 = T w^H V w``) -> per-bin gradient step.
"""
import numpy as np

from .overiva_oracle import input_covariance, projection_back, weighted_cov_all

EPS_R = 1e-15  # ive.py:203

UPDATES = ("demix", "mix", "switching")


def _H(T):
    return np.conj(T).swapaxes(1, 2)                     # ive.py:107-108


def _init(X, W0, init_eig):
    """prologue, ive.py:96-135: Cx, its inverse and norm, w (F, M, 1)"""
    T, F, M = X.shape
    Cx = input_covariance(X)                             # ive.py:100
    Cx_inv = np.linalg.inv(Cx)                           # ive.py:101
    Cx_norm = np.linalg.norm(Cx, axis=(1, 2))            # ive.py:102
    w = np.zeros((F, M, 1), dtype=X.dtype)
    if W0 is None:
        if init_eig:                                     # ive.py:111-126: principal eigenvector, NOT conjugated
            eigval, eigvec = np.linalg.eig(Cx)
            for f in range(F):
                w[f, :, 0] = eigvec[f, :, np.argmax(eigval[f])]
        else:
            w[:, 0] = 1.0                                # ive.py:129-130
    else:
        w[:, :] = W0                                     # ive.py:132-133
    return Cx, Cx_inv, Cx_norm, w


def _switching(a, Cx, Cx_norm, M):
    """ive.py:146-166: True where the mixing-vector step is to be used"""
    a_n = a / a[:, :1, :1]
    b_n = Cx @ a_n
    lmb = b_n[:, :1, :1].copy()
    b_n = b_n / lmb
    p1 = np.linalg.norm(a_n - b_n, axis=(1, 2)) / Cx_norm
    Cbb = lmb * (b_n @ _H(b_n)) / np.linalg.norm(b_n, axis=(1, 2), keepdims=True) ** 2
    p2 = np.linalg.norm(Cx - Cbb, axis=(1, 2))
    kappa = p1 * p2 / np.sqrt(M)
    return kappa >= 0.1


def ogive_faithful(X, n_iter=4000, step_size=0.1, tol=1e-3, update="demix", proj_back=True, W0=None,
                   model="laplace", init_eig=False, return_filters=False, callback=None, return_epochs=False):
    """ive.py:33-256 keeping its order of operations; ``return_epochs`` additionally returns how many epochs ran"""
    T, F, M = X.shape
    Cx, Cx_inv, Cx_norm, w = _init(X, W0, init_eig)
    a = np.zeros((F, M, 1), dtype=X.dtype)
    delta = np.zeros((F, M, 1), dtype=X.dtype)
    lambda_a = np.zeros((F, 1, 1), dtype=np.float64)

    def update_a_from_w(I):                              # ive.py:136-139
        v_new = Cx[I] @ w[I]
        lambda_w = 1.0 / np.real(_H(w[I]) @ v_new)
        a[I, :, :] = lambda_w * v_new

    def update_w_from_a(I):                              # ive.py:141-144 (lambda_a is refreshed for EVERY bin)
        v_new = Cx_inv @ a
        lambda_a[:] = 1.0 / np.real(_H(a) @ v_new)
        w[I, :, :] = lambda_a[I] * v_new[I]

    update_a_from_w(np.ones(F, dtype=bool))              # ive.py:173
    if update == "mix":                                  # ive.py:175-180
        I_do_w, I_do_a = np.zeros(F, dtype=bool), np.ones(F, dtype=bool)
    else:
        I_do_w, I_do_a = np.ones(F, dtype=bool), np.zeros(F, dtype=bool)

    r = np.zeros((T, 1))
    Y = np.zeros((F, T, 1), dtype=X.dtype)
    Xf = X.swapaxes(0, 1).copy()                         # ive.py:187-188
    epochs = 0
    for epoch in range(n_iter):                          # ive.py:190
        if update == "switching" and epoch % 10 == 0:    # ive.py:192-193
            I_do_a = _switching(a, Cx, Cx_norm, M)
            I_do_w = ~I_do_a
        Y[:, :, :] = Xf @ np.conj(w)                     # ive.py:196
        if callback is not None and epoch % 100 == 0:    # ive.py:199-205
            Yt = Y.swapaxes(0, 1)
            callback(Yt * np.conj(projection_back(Yt, X[:, :, 0])[None]) if proj_back else Yt)
        if model == "laplace":                           # ive.py:209-213
            r[:, :] = np.linalg.norm(Y, axis=0) / np.sqrt(F)
        elif model == "gauss":
            r[:, :] = (np.linalg.norm(Y, axis=0) ** 2) / F
        r[r < EPS_R] = EPS_R                             # ive.py:215-216
        r_inv = 1.0 / r
        psi = r_inv[None, :, :] * np.conj(Y)             # ive.py:221
        zeta = Y.swapaxes(1, 2) @ psi                    # ive.py:225
        x_psi = (Xf.swapaxes(1, 2) @ psi) / zeta         # ive.py:227
        delta[I_do_w] = a[I_do_w] - x_psi[I_do_w]        # ive.py:231-232
        w[I_do_w] += step_size * delta[I_do_w]
        delta[I_do_a] = w[I_do_a] - (Cx_inv[I_do_a] @ x_psi[I_do_a]) * lambda_a[I_do_a]   # ive.py:236-237
        a[I_do_a] += step_size * delta[I_do_a]
        update_a_from_w(I_do_w)                          # ive.py:240-241
        update_w_from_a(I_do_a)
        epochs = epoch + 1
        if np.max(np.linalg.norm(delta, axis=(1, 2))) < tol:   # ive.py:243-246
            break
    Y[:, :, :] = Xf @ np.conj(w)                         # ive.py:249
    Y = Y.swapaxes(0, 1).copy()
    if proj_back:                                        # ive.py:254-256
        Y *= np.conj(projection_back(Y, X[:, :, 0])[None])
    out = (Y, w) if return_filters else Y
    return (out, epochs) if return_epochs else out


# --------------------------------------------------------------------------
# staged form: the cuts the HIP kernels use (float64 / complex128 throughout)
# --------------------------------------------------------------------------
def ogive_activation(p, F, model):
    """r_inv (T, 1) from the summed power p = sum_f |y|^2   (ive.py:209-217)"""
    r = np.sqrt(p) / np.sqrt(F) if model == "laplace" else p / F
    return 1.0 / np.maximum(r, EPS_R)


def ogive_step_bin(w, a, delta, lambda_a, V, Cx, Cx_inv, do_a, step_size):
    """one epoch of the per-bin part (ive.py:221-241) given V = (1/T) sum_t r_inv x x^H of this epoch;
    all arguments (F, ...) arrays, updated in place; returns the per-bin norms of delta"""
    Vw = V @ w
    x_psi = Vw / (_H(w) @ Vw)                            # = (X^T psi) / zeta
    do_w = ~do_a
    delta[do_w] = a[do_w] - x_psi[do_w]
    w[do_w] += step_size * delta[do_w]
    delta[do_a] = w[do_a] - (Cx_inv[do_a] @ x_psi[do_a]) * lambda_a[do_a]
    a[do_a] += step_size * delta[do_a]
    v = Cx[do_w] @ w[do_w]
    a[do_w] = v / np.real(_H(w[do_w]) @ v)
    v = Cx_inv @ a
    lambda_a[:] = 1.0 / np.real(_H(a) @ v)
    w[do_a] = lambda_a[do_a] * v[do_a]
    return np.linalg.norm(delta, axis=(1, 2))


def ogive_staged(X, n_iter=4000, step_size=0.1, tol=1e-3, update="demix", proj_back=True, W0=None,
                 model="laplace", init_eig=False, return_filters=False, callback=None, return_epochs=False):
    T, F, M = X.shape
    Xc = X.astype(np.complex128)
    Cx, Cx_inv, Cx_norm, w = _init(Xc, W0, init_eig)
    v = Cx @ w
    a = v / np.real(_H(w) @ v)
    delta = np.zeros_like(a)
    lambda_a = np.zeros((F, 1, 1))
    do_a = np.ones(F, dtype=bool) if update == "mix" else np.zeros(F, dtype=bool)
    epochs = 0
    for epoch in range(n_iter):
        if update == "switching" and epoch % 10 == 0:
            do_a = _switching(a, Cx, Cx_norm, M)
        if callback is not None and epoch % 100 == 0:
            Yt = np.einsum("tfm,fmk->tfk", Xc, np.conj(w))
            callback((Yt * np.conj(projection_back(Yt, Xc[:, :, 0])[None]) if proj_back else Yt).astype(X.dtype))
        Y = np.einsum("tfm,fmk->tfk", Xc, np.conj(w))
        p = np.sum(np.abs(Y) ** 2, axis=1)
        V = weighted_cov_all(Xc, ogive_activation(p, F, model))[0]
        dn = ogive_step_bin(w, a, delta, lambda_a, V, Cx, Cx_inv, do_a, step_size)
        epochs = epoch + 1
        if dn.max() < tol:
            break
    Y = np.einsum("tfm,fmk->tfk", Xc, np.conj(w))
    if proj_back:
        Y = Y * np.conj(projection_back(Y, Xc[:, :, 0])[None])
    Y = Y.astype(X.dtype)
    out = (Y, w.astype(X.dtype)) if return_filters else Y
    return (out, epochs) if return_epochs else out
