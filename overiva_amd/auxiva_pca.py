"""Drop-in ``auxiva_pca()`` -- PCA to ``n_src`` channels followed by determined AuxIVA
(reference ``auxiva_pca.py:30-92``).

The covariance and the AuxIVA iterations run on the GPU; the Hermitian eigendecomposition of the
(n_freq, n_chan, n_chan) covariance stays on the host (LAPACK, as in the reference,
``auxiva_pca.py:75``) and so does the one-off projection onto the principal subspace.
"""
import numpy as np

from . import overiva as _ov
from .plan import Plan


def auxiva_pca(X, n_src=None, **kwargs):
    """
    Overdetermined IVA as PCA + determined AuxIVA.

    Same call contract as the reference, quirks included: ``proj_back`` must be passed (it is popped
    unconditionally, ``auxiva_pca.py:86``) and is then ignored -- the result is always projected
    back onto channel 0 of the ORIGINAL input (``auxiva_pca.py:89-90``); ``n_src`` is not forwarded,
    the inner solve is determined on the reduced channels (``auxiva_pca.py:87``); only ``Y`` is returned.
    """
    X = np.asarray(X)
    dtype = _ov._complex_dtype(X)
    n_frames, n_freq, n_chan = X.shape
    if n_src is None:
        n_src = n_chan
    kwargs.pop("proj_back")                                                   # auxiva_pca.py:86
    kwargs.pop("return_filters", None)   # the reference would hand a tuple to projection_back and fail

    precision = _ov.get_precision()
    with Plan(n_frames, n_freq, n_chan, n_src, "laplace", device=_ov.get_device()) as full:
        full.set_precision(precision)
        full.set_x(X)
        full.covariance()                                                     # auxiva_pca.py:71
        if n_src < n_chan:
            _, vecs = np.linalg.eigh(full.get_cx(np.complex128))               # auxiva_pca.py:75 (host LAPACK)
            P = np.ascontiguousarray(vecs[:, :, -n_src:])                     # (F, M, K) principal subspace
            full.set_w(P)
            new_X = full.demix(proj_back=False).astype(dtype, copy=False)     # x -> P^H x, auxiva_pca.py:79-81
        else:
            P = None
            new_X = X
        _, W_red = _ov.overiva(new_X, proj_back=False, return_filters=True, **kwargs)   # auxiva_pca.py:87
        # y = W_red^H (P^H x) = (P W_red)^H x : demix the ORIGINAL input with the composed filters and let
        # the epilogue kernel project back onto its channel 0                # auxiva_pca.py:89-90
        W_tot = W_red.astype(np.complex128) if P is None else np.matmul(P, W_red.astype(np.complex128))
        full.set_w(W_tot)
        Y = full.demix(proj_back=True).astype(dtype, copy=False)
    return Y
