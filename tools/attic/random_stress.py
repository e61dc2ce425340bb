#!/usr/bin/env python3
"""GPU box: N random shapes (2..16 channels, every source count, 1..80 bins, up to 700 frames, both models, both input dtypes)
through the drop-in call against the oracle, 3 iterations on i.i.d. input -- the bound of tests/test_gpu_parity.py's random-shape
test (1e-5, or 1.5 floors of the reference-faithful complex64 form where that is farther).  usage: random_stress.py [N] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import overiva_amd as oa
from oracle import overiva_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
bad = skipped = done = 0
worst = 0.0
for it in range(N):
    M = int(rng.integers(2, 17)); K = int(rng.integers(1, M + 1)); F = int(rng.integers(1, int(os.environ.get("FMAX", 80)))); T = int(rng.integers(4 * M, int(os.environ.get("TMAX", 700))))
    model = ("laplace", "gauss")[it % 2]
    X = orc.synth_iid(T, F, M, seed=5000 + it)
    with np.errstate(all="ignore"):
        try:
            Yr, Wr = orc.overiva_staged(X, n_src=K, n_iter=3, proj_back=True, model=model, return_filters=True)
        except np.linalg.LinAlgError:
            skipped += 1; continue
    if not (np.all(np.isfinite(Wr)) and np.all(np.isfinite(Yr))):
        skipped += 1; continue
    floor = None
    for dt in (np.complex64, np.complex128):
        try:
            Y, W = oa.overiva(X.astype(dt), n_src=K, n_iter=3, proj_back=True, model=model, return_filters=True)
            eW, eY = orc.rel_err(W, Wr), orc.rel_err(Y, Yr)
        except np.linalg.LinAlgError:
            eW = eY = np.inf
        bound = 1e-5
        if not (eW < bound and eY < bound):
            if floor is None:
                with np.errstate(all="ignore"):
                    try:
                        W64 = orc.overiva_faithful(X, n_src=K, n_iter=3, proj_back=True, model=model, return_filters=True)[1]
                        floor = orc.rel_err(W64, Wr) if np.all(np.isfinite(W64)) else np.inf
                    except np.linalg.LinAlgError:
                        floor = np.inf
            if not floor < 1e-2:
                skipped += 1; continue
            bound = max(1e-5, 1.5 * floor)
        done += 1
        worst = max(worst, eW / bound)
        if not (eW < bound and eY < 2 * bound):
            bad += 1
            print(f"FAIL ({T}, {F}, {M}, {K}) {model} {dt.__name__}: W {eW:.2e} Y {eY:.2e} bound {bound:.1e} floor {floor}", flush=True)
print(f"{done} comparisons, {bad} outside the bound, {skipped} skipped (degenerate / chaotic); worst W error / bound {worst:.2f}")
