#!/usr/bin/env python3
"""GPU box: configs[4] with float64 and with float32 partial blocks of the matrix-core covariance kernel ($OIVA_HMFMA_PART32, read when
a plan chooses its geometry): W after 1, 3 and 10 iterations on i.i.d. and on mixture-like input, relative difference between the two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, overiva_amd as oa
from oracle import overiva_oracle as orc
T, F, M, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (4000, 2048, 16, 16)
for kind in ("iid", "mixture"):
    X = orc.synth_iid(T, F, M, seed=2) if kind == "iid" else orc.synth_mixture(T, F, M, K, seed=3)
    W = {}
    for part in ("0", "1"):
        os.environ["OIVA_HMFMA_PART32"] = part
        with oa.Plan(T, F, M, K, "laplace") as p:
            p.set_precision("mixed"); p.set_x(X); p.covariance(); p.set_w(None)
            out = []
            for n in (1, 2, 7):
                p.iterate(n); out.append(p.get_w(np.complex128))
            W[part] = out
    for its, a, b in zip((1, 3, 10), W["0"], W["1"]):
        e = np.linalg.norm((b - a).reshape(F, -1), axis=1) / np.linalg.norm(a.reshape(F, -1), axis=1)
        print(f"{F}x{T}x{M}/{K} {kind:8s} after {its:2d} iterations: float32 vs float64 partial blocks: W rel {orc.rel_err(b, a):.2e} (per bin: median {np.median(e):.1e}, max {e.max():.1e})", flush=True)
