#!/usr/bin/env python3
"""GPU box: final W of a fixture row against the reference's complex128 / complex64 results for forced numbers of frame splits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import overiva_amd as oa
fid, model, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", f"overiva_{fid}.npz"))
X = d["X"]; T, F, M = X.shape; K = int(d["K"])
W128, W64 = d[f"W_c128_{model}_{n}"], d[f"W_c64_{model}_{n}"]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print(fid, model, n, "floor", rel(W64, W128))
for mode in ("mixed", "precise", "fast"):
    for ns in (0, 1, 2, 3, 4, 8):
        with oa.Plan(T, F, M, K, model) as p:
            p.set_precision(mode)
            if ns:
                p.set_cov_splits(ns)
            p.set_x(X); p.covariance(); p.set_w(None); p.iterate(n)
            W = p.get_w()
            print(f"  {mode} splits {ns}->{p.cov_splits()}: W vs c128 {rel(W, W128):.2e}  vs c64 {rel(W, W64):.2e}")
