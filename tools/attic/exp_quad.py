#!/usr/bin/env python3
"""GPU box: the four-lanes-per-(bin, frame) covariance kernel (csrc/kernels_cov_quad.hip) against the planar matrix-core
kernel on 10..16 channels with few sources: weighted covariance of both against the oracle on a small ragged shape, then
per-kernel times of the iteration at 2048 bins x 4000 frames.   python tools/exp_quad.py [--small-only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import overiva_amd as oa
from oracle import overiva_oracle as orc

for (T, F, M, K) in ((163, 19, 16, 2), (150, 18, 10, 2), (141, 17, 14, 1), (160, 16, 12, 3), (200, 33, 16, 4)):
    X = orc.synth_mixture(T, F, M, K, seed=4)
    rinv = np.random.default_rng(5).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    ref = orc.weighted_cov_all(X, rinv.astype(np.float64))
    for quad in (True, False):
        for splits in (0, 1, 3):
            with oa.Plan(T, F, M, K, "laplace") as p:
                p.set_precision("fast")
                active = p.set_cov_quad(quad)
                if splits:
                    p.set_cov_splits(splits)
                p.set_x(X)
                p.covariance()
                cx = orc.rel_err(p.get_cx(), orc.input_covariance(X.astype(np.complex128)))
                p.t_set_rinv(rinv)
                p.t_run_weighted_cov()
                V = p.t_get_v(np.complex128)
                print(f"{(T, F, M, K)} quad={active} splits={p.cov_splits()}: V err {orc.rel_err(V, ref):.2e}  Cx err {cx:.2e}  "
                      f"hermitian {np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))}", flush=True)

if "--small-only" in sys.argv:
    sys.exit(0)
T, F = 4000, 2048
rng = np.random.default_rng(0)
for M in (16, 12):
    X = (rng.standard_normal((T, F, M), dtype=np.float32) + 1j * rng.standard_normal((T, F, M), dtype=np.float32)).astype(np.complex64)
    for K in (1, 2, 3, 4):
        for mode in ("fast", "mixed"):
            with oa.Plan(T, F, M, K, "laplace") as p:
                p.set_precision(mode)
                p.set_x(X)
                p.covariance()
                for quad in (True, False):
                    active = p.set_cov_quad(quad)
                    p.set_w(None)
                    p.iterate(3)
                    best = None
                    for _ in range(3):
                        total, per = p.iterate_timed(10, per_kernel=True)
                        if best is None or total < best[0]:
                            best = (total, per)
                    total, per = best
                    print(f"{F}x{T}x{M}/{K} {mode} quad={active} splits={p.cov_splits()}: {total / 10 * 1e3:.1f} us/it  "
                          + "  ".join(f"{k} {v / 10 * 1e3:.1f}" for k, v in per.items()), flush=True)
    del X
