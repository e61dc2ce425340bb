#!/usr/bin/env python3
"""
How reproducible is the REFERENCE's own complex64 arithmetic?  Runs ONLY in the build container (needs /root/reference).

For every committed fixture ``overiva_<id>.npz`` x model x n_iter that holds the reference's complex64 result ``W_c64_*``, the
real ``/root/reference/overiva.py`` (imported as in make_golden.py) is run again in complex64 on the same X with every
sample moved by one unit in the last place (a relative perturbation of 2^-24 x N(0, 1), three seeds) and the largest
relative change of its W is stored:

    c64_jitter.npz:  "<id>_<model>_<n_iter>" -> max_seeds ||W64(X') - W64(X)||_F / ||W64(X)||_F

This is the complex64 counterpart of ``amp_*`` (which perturbs the complex128 run by 1e-12): a row whose jitter exceeds
1e-3 cannot be pinned on the reference's complex64 result any tighter than that jitter -- the reference's own answer moves
by that much when its input changes in the last bit (tests/conftest.py::c64_diverged, tests/test_gpu_parity.py).

Usage:  python tests/golden/make_jitter_golden.py
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402

SEEDS = (100, 101, 102)


def main():
    ref, _ = mg.import_reference()
    out = {}
    for path in sorted(glob.glob(os.path.join(HERE, "overiva_*.npz"))):
        fid = os.path.basename(path)[len("overiva_"):-len(".npz")]
        with np.load(path) as d:
            X, K = d["X"], int(d["K"])
            rows = [(m, n, d[f"W_c64_{m}_{n}"]) for m in ("laplace", "gauss") for n in (1, 2, 5, 20) if f"W_c64_{m}_{n}" in d.files]
        for model, n, W64 in rows:
            worst = 0.0
            for s in SEEDS:
                pert = (1 + 2.0 ** -24 * np.random.default_rng(s).standard_normal(X.shape)).astype(np.float32)
                Xp = (X * pert).astype(np.complex64)
                with np.errstate(all="ignore"):
                    try:
                        _, Wp = ref.overiva(Xp, n_src=K, n_iter=n, proj_back=False, model=model, return_filters=True)
                        e = np.linalg.norm(Wp - W64) / np.linalg.norm(W64)
                    except np.linalg.LinAlgError:
                        e = np.inf
                worst = max(worst, float(e) if np.isfinite(e) else np.inf)
            out[f"{fid}_{model}_{n}"] = np.float64(worst)
            if worst > 1e-3:
                print(f"{fid} {model} {n}: complex64 jitter of the reference {worst:.1e}")
    np.savez_compressed(os.path.join(HERE, "c64_jitter.npz"), **out)
    print(f"{len(out)} rows")


if __name__ == "__main__":
    main()
