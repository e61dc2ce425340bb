#!/usr/bin/env python3
"""GPU box: the covariance pass of configs[4] (2048 x 4000 x 16 / 16, mixed) against the frame splits, the sources on the
matrix cores (csrc/kernels_cov_hmfma.hip) and on the vector ALU alone (csrc/kernels_cov_half16.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (4000, 2048, 16, 16)
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
for on in (True, False):
    p.set_cov_hmfma(on)
    for ns in (1, 2, 4, 8, 16):
        p.set_cov_splits(ns); p.iterate(1); p.sync()
        tc = min(p.t_time_stage('weighted_cov', 10) * 1e3 for _ in range(3)); tu = min(p.t_time_stage('ip_update', 10) * 1e3 for _ in range(3))
        print(f"matrix cores {on!s:5s} splits {p.cov_splits():3d}: cov {tc:7.1f} us, update {tu:6.1f} us", flush=True)
