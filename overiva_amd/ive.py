"""Drop-in ``ogive()`` -- orthogonally constrained independent vector extraction of one source (Koldovsky &
Tichavsky 2018), same name, argument order, defaults and return values as reference ``ive.py:33-45``; the body runs
on one MI355X.  Per epoch the GPU does the demix + power pass, the activation, the weighted covariance of the single
source (``x_psi = V w / (w^H V w)``, see csrc/kernels_ogive.hip) and the per-bin gradient step with its orthogonal
constraints in float64; the stopping rule (``ive.py:243-246``) is evaluated on the device after every epoch, the host
only looks every ``CHUNK`` epochs (a state that met the rule is frozen, so the result is that of the reference's
``break``).

Same arithmetic modes and deviations as ``overiva()`` (``overiva_amd/overiva.py``); the callback fires every 100
epochs (``ive.py:199``).  The MATLAB wrapper of the reference (``ive.py:259-428``) is not reproduced.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import overiva as _ov
from .plan import Plan

CHUNK = 50
UPDATE_IDS = {"demix": 0, "mix": 1, "switching": 2}


def ogive(
    X,
    n_iter=4000,
    step_size=0.1,
    tol=1e-3,
    update="demix",
    proj_back=True,
    W0=None,
    model="laplace",
    init_eig=False,
    return_filters=False,
    callback=None,
):
    """
    Orthogonally constrained independent vector extraction (OGIVE) on the GPU.

    Parameters
    ----------
    X: ndarray (nframes, nfrequencies, nchannels)
        STFT representation of the signal
    n_iter: int, optional
        Maximum number of gradient steps (default 4000)
    step_size: float
        Step size of the gradient ascent
    tol: float
        Stop when the largest per-bin step is smaller than this
    update: str
        'demix' (default), 'mix' or 'switching'
    proj_back: bool, optional
        Scale by projection back onto the first channel (default True)
    W0: ndarray broadcastable to (nfrequencies, nchannels, 1), optional
        Initial demixing vector
    model: str
        'laplace' (default) or 'gauss'
    init_eig: bool, optional
        Start from the principal eigenvector of the input covariance when ``W0 is None``
    return_filters: bool
        Also return the demixing vector (nfrequencies, nchannels, 1)
    callback: func
        Called with the current (nframes, nfrequencies, 1) estimate every 100 epochs

    Returns
    -------
    Y (nframes, nfrequencies, 1) in the dtype of X, or ``(Y, w)`` if ``return_filters``.
    """
    X = np.asarray(X)
    if X.ndim != 3:
        raise ValueError("X must have shape (n_frames, n_freq, n_chan)")
    dtype = _ov._complex_dtype(X)
    n_frames, n_freq, n_chan = X.shape
    if model not in ("laplace", "gauss"):
        raise ValueError(f"model must be 'laplace' or 'gauss', got {model!r}")
    if update not in UPDATE_IDS:                 # the reference silently treats anything else as 'demix' (ive.py:175-180)
        raise ValueError(f"update must be one of {sorted(UPDATE_IDS)}, got {update!r}")
    if n_iter < 0:
        raise ValueError("n_iter must be >= 0")
    precision = _ov.get_precision()
    if precision == "auto":          # the per-bin gradient step is float64 whatever the mode; OGIVE was validated with the float64 covariance pass
        precision = "precise"
    wdtype = np.complex64 if precision == "fast" else np.complex128
    with Plan(n_frames, n_freq, n_chan, 1, model, device=_ov.get_device()) as plan:
        plan.set_precision(precision)
        plan.set_x(X)
        plan.covariance()                                                   # ive.py:100
        if W0 is None and init_eig:                                         # ive.py:111-126 (host LAPACK; not conjugated)
            vals, vecs = np.linalg.eig(plan.get_cx(np.complex128))
            W0 = np.stack([vecs[f][:, np.argmax(vals[f])] for f in range(n_freq)])[:, :, None]
        plan.set_w(None if W0 is None else np.broadcast_to(np.asarray(W0), (n_freq, n_chan, 1)))
        plan.ogive_begin(update, model)
        epoch, converged = 0, False
        while epoch < n_iter and not converged:
            if callback is not None and epoch % 100 == 0:                  # ive.py:199-205
                callback(plan.demix(proj_back).astype(dtype, copy=False))
            step = min(n_iter - epoch, CHUNK)
            if callback is not None:
                step = min(step, 100 - epoch % 100)
            ran, converged, _ = plan.ogive_iterate(epoch, step, step_size, tol)
            epoch += ran if converged else step
        Y = plan.demix(proj_back).astype(dtype, copy=False)                 # ive.py:249-256
        w = plan.get_w(wdtype)                                              # raises LinAlgError on a non-finite w
    if return_filters:
        return Y, w.astype(dtype, copy=False)
    return Y
