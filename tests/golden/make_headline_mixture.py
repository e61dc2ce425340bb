#!/usr/bin/env python3
"""Expected results of tests/test_gpu_parity.py::test_headline_mixture_20_iterations, computed ONCE by the oracle (about two
minutes of CPU work per GPU test run otherwise): 2048 bins x 4000 frames x 8 mics / 2 sources, mixture-like input of seed
21, 20 iterations -- W of the reference-faithful complex64 form and of the complex128 form.  These are ORACLE outputs, not
reference outputs (the reference needs 20 x 6 s per iteration and 8 GB of temporaries at this size); the oracle itself is
pinned on the reference's own golden vectors by tests/test_oracle_golden.py.  The test recomputes them when the file is
absent or when the checksum of the regenerated input differs.     python tests/golden/make_headline_mixture.py"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import overiva_oracle as orc

SHAPE, SEED, N_ITER = (4000, 2048, 8, 2), 21, 20
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "headline_mixture20.npz")


def x_digest(X):
    return hashlib.sha256(np.ascontiguousarray(X[::97, ::31]).tobytes()).hexdigest()


def compute(X, K):
    _, W64 = orc.overiva_faithful(X, n_src=K, n_iter=N_ITER, proj_back=False, return_filters=True)
    _, W128 = orc.overiva_staged(X.astype(np.complex128), n_src=K, n_iter=N_ITER, proj_back=False, return_filters=True)
    return W64, W128


if __name__ == "__main__":
    T, F, M, K = SHAPE
    X = orc.synth_mixture(T, F, M, K, seed=SEED)
    W64, W128 = compute(X, K)
    np.savez_compressed(OUT, W64=W64, W128=W128, x_digest=x_digest(X), shape=np.array(SHAPE), seed=SEED, n_iter=N_ITER)
    print(OUT, os.path.getsize(OUT), "bytes; floor", orc.rel_err(W64, W128))
