#!/usr/bin/env python3
"""GPU box: one case of tests/test_update16r_gpu.py, per-bin difference between the two update kernels (run twice with
OIVA_DET16_ROWS=0 / 1; the second run compares with the first one's file)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, overiva_amd as oa
from oracle import overiva_oracle as orc
T, F, M, splits = [int(a) for a in sys.argv[1:5]]
X = orc.synth_mixture(T, F, M, M, seed=T + F + M) if F % 2 else orc.synth_iid(T, F, M, seed=T + F + M)
with oa.Plan(T, F, M, M, "laplace") as p:
    p.set_precision("mixed"); p.set_x(X); p.covariance(); p.set_w(None)
    if splits: p.set_cov_splits(splits)
    Ws = []
    for it in range(3):
        p.iterate(1); Ws.append(p.get_w(np.complex128))
    V = p.t_get_v(np.complex128)
tag = os.environ.get("OIVA_DET16_ROWS", "1")
np.savez(f"/tmp/case_{tag}.npz", W=np.array(Ws))
other = f"/tmp/case_{'0' if tag == '1' else '1'}.npz"
if os.path.exists(other):
    Wo = np.load(other)["W"]
    for it in range(3):
        e = np.linalg.norm((Ws[it] - Wo[it]).reshape(F, -1), axis=1) / np.linalg.norm(Wo[it].reshape(F, -1), axis=1)
        worst = np.argsort(e)[-3:][::-1]
        print(f"iteration {it + 1}: total {orc.rel_err(Ws[it], Wo[it]):.2e}; worst bins {[(int(b), float(f'{e[b]:.1e}')) for b in worst]}; median {np.median(e):.1e}")
    cond = np.array([[np.linalg.cond(V[s, f]) for s in range(M)] for f in range(F)])
    print("cond(V_s) of the worst bin of the last iteration:", [float(f"{c:.1e}") for c in cond[worst[0]]][:8], " median over all:", f"{np.median(cond):.1e}", " max:", f"{cond.max():.1e}")
