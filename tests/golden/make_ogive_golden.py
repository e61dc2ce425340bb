#!/usr/bin/env python3
"""
Golden-vector generator for OGIVE.  Runs ONLY in the build container (needs /root/reference).

Imports the real reference ``/root/reference/ive.py`` (unmodified, read-only) and records inputs and outputs of
``ogive()`` (ive.py:33-256) as one small ``.npz`` next to this file.  Shims needed in this image: the
``pyroomacoustics`` stub of make_golden.py (``ive.py:30`` imports ``projection_back``) and ``np.bool``, which
``ive.py:173-180`` uses and NumPy >= 1.24 removed -- supplied through a module-global proxy, the file is untouched.

Usage:  python tests/golden/make_ogive_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle.overiva_oracle import projection_back, synth_mixture  # noqa: E402


def import_reference():
    pra = types.ModuleType("pyroomacoustics")
    bss = types.ModuleType("pyroomacoustics.bss")
    bss.projection_back = projection_back
    pra.bss = bss
    sys.modules["pyroomacoustics"] = pra
    sys.modules["pyroomacoustics.bss"] = bss
    sys.path.insert(0, "/root/reference")
    import ive

    class _Np:
        bool = bool

        def __getattr__(self, name):
            return getattr(np, name)

    ive.np = _Np()
    return ive


CASES = [("a", 60, 9, 4), ("b", 80, 12, 6), ("c", 64, 5, 2), ("d", 96, 16, 8)]


def main():
    ive = import_reference()
    out = {}
    for name, T, F, M in CASES:
        X = synth_mixture(T, F, M, 2, seed=300 + ord(name)).astype(np.complex128)
        out[f"{name}_X"] = X.astype(np.complex64)          # inputs are exactly representable in complex64
        X = out[f"{name}_X"].astype(np.complex128)
        pert = 1.0 + 1e-12 * np.random.default_rng(7).standard_normal(X.shape)
        for update in ("demix", "mix", "switching"):
            for model in ("laplace", "gauss"):
                for n_iter in (1, 5, 20, 200):
                    Y, w = ive.ogive(X.copy(), n_iter=n_iter, tol=0.0, update=update, proj_back=False, model=model,
                                     return_filters=True)
                    key = f"{name}_{update}_{model}_{n_iter}"
                    if not np.all(np.isfinite(w)):
                        continue
                    out[f"W_{key}"] = np.array(w)
                    _, wp = ive.ogive(X * pert, n_iter=n_iter, tol=0.0, update=update, proj_back=False, model=model,
                                      return_filters=True)
                    out[f"amp_{key}"] = np.float64(np.linalg.norm(wp - w) / np.linalg.norm(w) / 1e-12)
                    # the reference's own complex64 run: its distance from the complex128 run is the noise floor of
                    # float32 arithmetic inside the loop, which a one-off input perturbation (amp) does not measure
                    with np.errstate(all="ignore"):
                        _, w64 = ive.ogive(X.astype(np.complex64), n_iter=n_iter, tol=0.0, update=update, proj_back=False,
                                           model=model, return_filters=True)
                    if np.all(np.isfinite(w64)):
                        out[f"floor_{key}"] = np.float64(np.linalg.norm(w64 - w) / np.linalg.norm(w))
        # default tolerance: the loop leaves early (ive.py:243-246); record how many epochs ran through the callback cadence
        got = []
        Y = ive.ogive(X.copy(), n_iter=400, tol=2e-2, proj_back=True, callback=lambda y: got.append(np.array(y)))
        out[f"{name}_Ytol"] = Y
        out[f"{name}_ncb"] = len(got)
        out[f"{name}_cb0"] = got[0]
        Y, w = ive.ogive(X.copy(), n_iter=30, proj_back=False, init_eig=True, return_filters=True)
        out[f"{name}_Weig"] = np.array(w)
        rng = np.random.default_rng(11)
        W0 = np.zeros((F, M, 1), complex)
        W0[:, 0] = 1.0
        W0 += 0.1 * (rng.standard_normal((F, M, 1)) + 1j * rng.standard_normal((F, M, 1)))
        out[f"{name}_W0"] = W0
        Y, w = ive.ogive(X.copy(), n_iter=30, proj_back=False, W0=W0, return_filters=True)
        out[f"{name}_Ww0"] = np.array(w)
    path = os.path.join(HERE, "ogive.npz")
    np.savez_compressed(path, **out)
    print(path, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")
    print({k: round(float(v), 1) for k, v in out.items() if k.startswith("amp_") and k.endswith("_200")})


if __name__ == "__main__":
    main()
