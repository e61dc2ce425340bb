import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, overiva_amd as oa
from oracle import overiva_oracle as orc
bad = 0
for M, K in ((16, 16), (16, 9), (15, 15), (16, 12), (12, 12), (14, 10)):
    for T in (16, 17, 20, 31, 32, 33, 40, 63, 64, 65, 100):
        for F in (1, 3):
            X = orc.synth_iid(T, F, M, seed=T + F + M)
            rinv = np.random.default_rng(T).gamma(2.0, 1.0, (T, K)).astype(np.float32)
            with oa.Plan(T, F, M, K, "laplace") as p:
                p.set_precision("mixed"); p.set_x(X); p.covariance(); p.t_set_rinv(rinv); p.t_run_weighted_cov(); V = p.t_get_v(np.complex128)
            ref = orc.weighted_cov_all(X, rinv.astype(np.float64))
            e = orc.rel_err(V, ref)
            if not e < 3e-6:
                bad += 1; print("FAIL", M, K, T, F, e)
print("bad", bad)
