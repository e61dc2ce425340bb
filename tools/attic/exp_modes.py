#!/usr/bin/env python3
"""GPU box: accuracy of the arithmetic modes in units of the reference's own complex64 floor.

For every mixture fixture x model x n_iter: final W of {fast, upd64 (float32 covariance + float64 per-bin algebra),
precise} against the reference's complex128 result, divided by the distance between the reference's complex64 and
complex128 results (the floor).  Then the same on a long-frame-axis mixture (256 x 4000 x 8 / 2, floor from the oracle's
reference-faithful complex64 form) for several covariance split counts (= float32 chain lengths)."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import overiva_amd as oa
from overiva_amd import _lib
from oracle import overiva_oracle as orc

MODES = (("fast", _lib.PREC_FAST), ("upd64", _lib.PREC_UPDATE_F64), ("precise", _lib.PREC_PRECISE))


def run(X, K, model, n_iter, flags, splits=0):
    T, F, M = X.shape
    with oa.Plan(T, F, M, K, model) as p:
        p.set_precision(flags)
        if splits:
            p.set_cov_splits(splits)
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.iterate(n_iter)
        return p.get_w(np.complex128)


worst = {m: 0.0 for m, _ in MODES}
ONLY = os.environ.get("OIVA_EXP_FIXTURES")          # e.g. "l,o,p,q": only these fixtures, and no long-frame-axis part
for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "overiva_*_mix.npz"))):
    if ONLY and os.path.basename(path)[8:-8] not in ONLY.split(","):
        continue
    g = np.load(path)
    X, K = g["X"], int(g["K"])
    name = os.path.basename(path)[8:-4]
    for model in ("laplace", "gauss"):
        for n in (1, 5, 20):
            k128, k64 = f"W_c128_{model}_{n}", f"W_c64_{model}_{n}"
            if k128 not in g.files or k64 not in g.files:
                continue
            amp = float(g[f"amp_{model}_{n}"]) if f"amp_{model}_{n}" in g.files else 1.0
            if amp > 1e3:
                continue
            ref = g[k128]
            floor = orc.rel_err(g[k64], ref)
            row = []
            for mname, flags in MODES:
                e = orc.rel_err(run(X.astype(np.complex64), K, model, n, flags), ref)
                fl = e / max(floor, 1e-30)
                if floor > 2e-7:
                    worst[mname] = max(worst[mname], fl)
                row.append(f"{mname} {e:.1e} ({fl:.2f} fl)")
            print(f"{name:6s} {model:7s} n={n:2d} amp {amp:6.1f} floor {floor:.1e} | " + " | ".join(row), flush=True)
print("worst (floors, where floor > 2e-7):", {k: round(v, 2) for k, v in worst.items()}, flush=True)

if ONLY:
    sys.exit(0)
# long frame axis: the float32 chains of the covariance pass are 62 frames at 4 splits
T, F, M, K = 4000, 256, 8, 2
for seed in (3, 4):
    X = orc.synth_mixture(T, F, M, K, seed=seed)
    for model in ("laplace", "gauss"):
        n = 20
        ref = orc.overiva_staged(X.astype(np.complex128), n_src=K, n_iter=n, proj_back=False, model=model, return_filters=True)[1]
        w64 = orc.overiva_faithful(X.astype(np.complex64), n_src=K, n_iter=n, proj_back=False, model=model, return_filters=True)[1]
        floor = orc.rel_err(w64, ref)
        row = []
        for mname, flags in MODES:
            for splits in ((0, 8, 16, 32) if mname != "precise" else (0,)):
                e = orc.rel_err(run(X, K, model, n, flags, splits), ref)
                row.append(f"{mname}/s{splits} {e:.1e} ({e / floor:.2f})")
        print(f"256x4000x8 seed {seed} {model} n=20 floor {floor:.1e} | " + " | ".join(row), flush=True)
