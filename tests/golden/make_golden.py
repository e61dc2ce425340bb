#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

Imports the real reference (``/root/reference/overiva.py`` / ``auxiva_pca.py``,
unmodified, read-only) and records inputs and outputs of its hot path as small
``.npz`` fixtures next to this file.  Nothing here travels as code to the GPU box
except this script itself; the fixtures are data (inputs + expected outputs).

Two shims are needed to import the reference in this image:

1. ``pyroomacoustics`` is not installed and ``overiva.py:25`` imports
   ``projection_back`` from it.  A stub module provides the restated formula
   (``oracle/overiva_oracle.py::projection_back`` -- parity unpinned, see there).
   With ``proj_back=False`` the stub is never called.
2. ``overiva.py:182`` calls ``np.linalg.solve(A, b)`` with a stack-of-vectors ``b``
   which NumPy >= 2 rejects; the module-global ``np`` of the imported reference is
   rebound to a proxy that restores the NumPy-1 behaviour.  The file is untouched.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle.overiva_oracle import projection_back, synth_iid, synth_mixture  # noqa: E402

REFERENCE = "/root/reference"


def import_reference():
    pra = types.ModuleType("pyroomacoustics")
    bss = types.ModuleType("pyroomacoustics.bss")
    bss.projection_back = projection_back
    pra.bss = bss
    sys.modules["pyroomacoustics"] = pra
    sys.modules["pyroomacoustics.bss"] = bss
    sys.path.insert(0, REFERENCE)
    import overiva as ref_overiva
    import auxiva_pca as ref_pca

    class _Linalg:
        def __getattr__(self, name):
            return getattr(np.linalg, name)

        @staticmethod
        def solve(a, b):
            a = np.asarray(a)
            b = np.asarray(b)
            if b.ndim == a.ndim - 1:
                return np.linalg.solve(a, b[..., None])[..., 0]
            return np.linalg.solve(a, b)

    class _Np:
        linalg = _Linalg()

        def __getattr__(self, name):
            return getattr(np, name)

    ref_overiva.np = _Np()
    return ref_overiva, ref_pca


# (name, T, F, M, K)
CASES = [
    ("a", 40, 5, 3, 1),
    ("b", 50, 9, 4, 2),
    ("c", 64, 11, 4, 4),
    ("d", 80, 20, 8, 2),
    ("e", 64, 4, 16, 16),
    ("f", 48, 6, 2, 2),
    ("g", 72, 7, 6, 3),
]
N_ITERS = (0, 1, 2, 5, 20)

# Larger, better conditioned shapes (F >= 64 bins, T >= 200 frames): with this many bins the gauss model's
# activation r = sum_f |y|^2 / F no longer gets arbitrarily close to 0, so 20 iterations of it can be pinned
# instead of skipped (amp_* stays small).  Only W is stored (Y = demix(X, W) follows from it).
BIG_CASES = [
    ("h", 200, 64, 4, 2),
    ("i", 200, 64, 8, 2),
    ("j", 208, 64, 6, 3),
    # channel counts without a small fixture: odd counts (generic covariance kernel), determined 8 x 8, and the padded
    # 9..16-channel path with background channels
    ("k", 160, 24, 7, 2),
    ("l", 160, 16, 12, 3),
    ("m", 144, 24, 8, 8),
    ("n", 128, 20, 5, 5),
    # 10 / 12 / 14 / 16 channels with few sources: the four-lanes-per-(bin, frame) covariance kernel (l above is 12 / 3);
    # ragged bin groups (F % 16 != 0) and frame counts that are no multiple of the 8-frame step
    ("o", 163, 19, 16, 2),
    ("p", 150, 18, 10, 2),
    ("q", 141, 17, 14, 1),
    # 9 / 11 / 13 / 15 channels: the same kernels on a copy of X padded by one zero channel; few and many sources, determined
    ("r", 157, 17, 9, 2),
    ("s", 150, 15, 13, 3),
    ("t", 144, 12, 15, 15),
    ("u", 133, 14, 11, 6),
    # (round 4) 6 and 2 channels with 1-2 sources + background: the X-resident kernel beyond 4 / 8 channels
    ("w", 160, 24, 6, 2),
    ("x", 150, 40, 6, 1),
    ("y", 176, 33, 2, 1),
    # (round 5) determined 16 x 16 and 15 x 15 from MORE frames than e (64) and t (144): the matrix-core covariance kernel of
    # configs[4] where the reference's own complex64 run stays reproducible for 20 iterations (tests/golden/c64_jitter.npz)
    ("ea", 256, 4, 16, 16),
    ("eb", 1024, 4, 16, 16),
    ("ta", 256, 5, 15, 15),
    ("tb", 1024, 3, 15, 15),
    # (round 5) 8 channels with 3 and 4 sources + background at the reference's own frame count: the four-sources-per-pass
    # covariance kernel (cov_pair32_kernel) and the Gram form of the update had no fixture from the real reference
    ("v", 235, 33, 8, 4),
    ("z", 160, 40, 8, 3),
]
BIG_ITERS = (1, 5, 20)


def make_input(family, T, F, M, K, seed):
    if family == "iid":
        return synth_iid(T, F, M, seed=seed)
    return synth_mixture(T, F, M, max(K, 1), seed=seed)


def trace_intermediates(ref_overiva, X, K, n_iter, model):
    """Snapshot V, r_inv, W_hat at every (epoch, s) right after overiva.py:179 executed
    (i.e. when line 181 is about to run), and W_hat at the end of each source update."""
    snaps = []
    target = ref_overiva.overiva.__code__

    def tracer(frame, event, arg):
        if frame.f_code is not target:
            return None

        def local(frame, event, arg):
            if event == "line" and frame.f_lineno == 181:
                loc = frame.f_locals
                snaps.append(dict(epoch=loc["epoch"], s=loc["s"], V=loc["V"].copy(),
                                  r_inv=loc["r_inv"].copy(), W_hat=loc["W_hat"].copy()))
            return local

        return local

    sys.settrace(tracer)
    try:
        Y, W = ref_overiva.overiva(X, n_src=K, n_iter=n_iter, proj_back=False, model=model,
                                   return_filters=True)
    finally:
        sys.settrace(None)
    return snaps, Y, np.array(W)


def main():
    ref_overiva, ref_pca = import_reference()
    if "--big-only" in sys.argv:
        make_big(ref_overiva, ref_pca)
        return
    total = 0
    for name, T, F, M, K in CASES:
        for family in ("iid", "mix"):
            seed = 1000 + 7 * len(name) + ord(name) + (0 if family == "iid" else 500)
            X64 = make_input(family, T, F, M, K, seed)
            out = {"X": X64, "T": T, "F": F, "M": M, "K": K}
            nonfinite = []
            pert = 1.0 + 1e-12 * np.random.default_rng(seed + 2).standard_normal(X64.shape)
            for dt_name, X in (("c64", X64), ("c128", X64.astype(np.complex128))):
                for model in ("laplace", "gauss"):
                    for n_iter in N_ITERS:
                        Y, W = ref_overiva.overiva(X.copy(), n_src=K, n_iter=n_iter, proj_back=False,
                                                   model=model, return_filters=True)
                        key = f"{dt_name}_{model}_{n_iter}"
                        assert Y.dtype == X.dtype and W.dtype == X.dtype
                        if not (np.all(np.isfinite(W)) and np.all(np.isfinite(Y))):
                            # the reference itself diverged (seen for complex64 / gauss /
                            # determined mixtures): nothing to pin, remember the key
                            nonfinite.append(key)
                            continue
                        out[f"W_{key}"] = np.ascontiguousarray(W)
                        if dt_name == "c128":
                            # conditioning of the reference itself: relative change of its W under a
                            # 1e-12 relative perturbation of X, divided by 1e-12
                            _, Wp = ref_overiva.overiva(X * pert, n_src=K, n_iter=n_iter, proj_back=False,
                                                        model=model, return_filters=True)
                            amp = np.linalg.norm(Wp - W) / np.linalg.norm(W) / 1e-12
                            out[f"amp_{model}_{n_iter}"] = np.float64(amp)
                        if n_iter == 20 and (dt_name == "c128" or model == "laplace"):
                            out[f"Y_{key}"] = Y
                    # proj_back=True epilogue + callback payloads (epochs 0 and 10)
                    if dt_name == "c64" and model == "gauss":
                        continue
                    got = []
                    Ypb = ref_overiva.overiva(X.copy(), n_src=K, n_iter=12, proj_back=True, model=model,
                                              callback=lambda y: got.append(np.array(y)))
                    out[f"Ypb_{dt_name}_{model}_12"] = Ypb
                    if dt_name == "c128":
                        Yp = ref_overiva.overiva(X * pert, n_src=K, n_iter=12, proj_back=True, model=model)
                        out[f"amp_{model}_12"] = np.float64(np.linalg.norm(Yp - Ypb) / np.linalg.norm(Ypb) / 1e-12)
                    if dt_name == "c128" and model == "laplace":
                        out[f"cb0_{dt_name}_{model}"] = got[0]
                        out[f"cb10_{dt_name}_{model}"] = got[1]
                    assert len(got) == 2
            X128 = X64.astype(np.complex128)
            # warm start: W0 broadcasting to (F, M, K)  (overiva.py:116-117)
            rng = np.random.default_rng(seed + 1)
            W0 = (np.eye(M, K)[None] + 0.1 * (rng.standard_normal((F, M, K))
                                               + 1j * rng.standard_normal((F, M, K))))
            out["W0"] = W0
            Y, W = ref_overiva.overiva(X128.copy(), n_src=K, n_iter=3, proj_back=False, W0=W0,
                                       return_filters=True)
            out["W_w0_c128_laplace_3"] = np.ascontiguousarray(W)
            # init_eig (compare up to per-column phase: eigvec phase is LAPACK-dependent)
            Y, W = ref_overiva.overiva(X128.copy(), n_src=K, n_iter=3, proj_back=False, init_eig=True,
                                       return_filters=True)
            out["W_eig_c128_laplace_3"] = np.ascontiguousarray(W)
            out["Y_eig_c128_laplace_3"] = Y
            # default n_src (determined AuxIVA on all channels)
            Y, W = ref_overiva.overiva(X128.copy(), n_iter=2, proj_back=False, return_filters=True)
            out["W_det_c128_laplace_2"] = np.ascontiguousarray(W)
            # auxiva_pca (auxiva_pca.py:63-92); needs proj_back kw, returns Y only
            Ypca = ref_pca.auxiva_pca(X128.copy(), n_src=K, n_iter=5, proj_back=True, model="laplace")
            out["Ypca_c128_laplace_5"] = Ypca
            # per-(epoch, source) intermediates, c128, two epochs
            if name in ("b", "d", "g", "c"):
                for model in ("laplace", "gauss"):
                    snaps, Y, W = trace_intermediates(ref_overiva, X128.copy(), K, 2, model)
                    assert len(snaps) == 2 * K
                    for sn in snaps:
                        tag = f"im_{model}_e{sn['epoch']}_s{sn['s']}"
                        out[f"{tag}_V"] = sn["V"]
                        out[f"{tag}_rinv"] = sn["r_inv"]
                        out[f"{tag}_What"] = sn["W_hat"]
                    out[f"im_{model}_Wfinal"] = W
            out["nonfinite"] = np.array(nonfinite, dtype="U32")
            path = os.path.join(HERE, f"overiva_{name}_{family}.npz")
            np.savez_compressed(path, **out)
            sz = os.path.getsize(path)
            total += sz
            print(f"{path}: {len(out)} arrays, {sz / 1024:.0f} KiB")
    total += make_big(ref_overiva, ref_pca)
    print(f"total {total / 1e6:.2f} MB")


def make_big(ref_overiva, ref_pca):
    total = 0
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]      # e.g. --big-only --only=o,p
    for name, T, F, M, K in BIG_CASES:
        if only and name not in only[0]:
            continue
        for family in ("iid", "mix"):
            seed = 2000 + sum(ord(c) for c in name) + (0 if family == "iid" else 500)
            X64 = make_input(family, T, F, M, K, seed)
            out = {"X": X64, "T": T, "F": F, "M": M, "K": K}
            nonfinite = []
            pert = 1.0 + 1e-12 * np.random.default_rng(seed + 2).standard_normal(X64.shape)
            for dt_name, X in (("c64", X64), ("c128", X64.astype(np.complex128))):
                for model in ("laplace", "gauss"):
                    for n_iter in BIG_ITERS:
                        Y, W = ref_overiva.overiva(X.copy(), n_src=K, n_iter=n_iter, proj_back=False, model=model,
                                                   return_filters=True)
                        key = f"{dt_name}_{model}_{n_iter}"
                        if not (np.all(np.isfinite(W)) and np.all(np.isfinite(Y))):
                            nonfinite.append(key)
                            continue
                        out[f"W_{key}"] = np.ascontiguousarray(W)
                        if dt_name == "c128":
                            _, Wp = ref_overiva.overiva(X * pert, n_src=K, n_iter=n_iter, proj_back=False, model=model,
                                                        return_filters=True)
                            out[f"amp_{model}_{n_iter}"] = np.float64(np.linalg.norm(Wp - W) / np.linalg.norm(W) / 1e-12)
            X128 = X64.astype(np.complex128)
            # proj_back epilogue: per-(bin, source) scale z, stored through W-sized quantities only: Y[0] rows
            Ypb = ref_overiva.overiva(X128.copy(), n_src=K, n_iter=12, proj_back=True, model="laplace")
            out["Ypb_frame0_c128_laplace_12"] = Ypb[0]                      # (F, K): pins z without storing (T, F, K)
            Ypca = ref_pca.auxiva_pca(X128.copy(), n_src=K, n_iter=5, proj_back=True, model="laplace")
            out["Ypca_frame0_c128_laplace_5"] = Ypca[0]
            out["nonfinite"] = np.array(nonfinite, dtype="U32")
            path = os.path.join(HERE, f"overiva_{name}_{family}.npz")
            np.savez_compressed(path, **out)
            sz = os.path.getsize(path)
            total += sz
            print(f"{path}: {len(out)} arrays, {sz / 1024:.0f} KiB; amps",
                  {k: round(float(v), 1) for k, v in out.items() if k.startswith("amp_")})
    return total


if __name__ == "__main__":
    main()
