"""Drop-in ``auxiva_pca()`` -- PCA to ``n_src`` channels followed by determined AuxIVA
(reference ``auxiva_pca.py:30-92``).

Everything runs on the GPU: the covariance, the Hermitian eigendecomposition (a Jacobi eigensolver, one wavefront per
bin, ``csrc/kernels_evd.hip``; the reference calls LAPACK, ``auxiva_pca.py:75``), the projection onto the principal
subspace -- whose result stays in device memory and is the input of the determined solve -- the AuxIVA iterations and
the final demix + projection back.  The eigenvectors' phases are the eigensolver's, not LAPACK's; the result does not
depend on them (a phase of a principal component becomes a phase of a demixed source, which projection back removes).
"""
import numpy as np

from . import overiva as _ov
from . import sharded
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

    precision = _ov.resolve_precision(dtype, n_chan, n_frames=n_frames)
    with Plan(n_frames, n_freq, n_chan, n_src, "laplace", device=_ov.get_device()) as full:
        full.set_precision(precision)
        full.set_x(X)
        full.covariance()                                                     # auxiva_pca.py:71
        if n_src < n_chan:
            full.set_w_pca()                                                  # auxiva_pca.py:75: eigh, w[:, :, -n_src:]
            P = full.get_w(np.complex128)                                     # (F, M, K) principal subspace
            if sharded.active_group() is not None:                            # a device-resident tensor cannot be sharded over ranks
                new_X = full.demix(proj_back=False, dtype=dtype)
            else:
                new_X = full.demix_device(proj_back=False)                    # x -> P^H x, auxiva_pca.py:79-81
                new_X.dtype = np.dtype(dtype)
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
