#!/usr/bin/env python3
"""GPU box: weighted-covariance pass time against the number of frame splits:  T F M K mode splits..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]]
mode = sys.argv[5]
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision(mode); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
for ns in [int(a) for a in sys.argv[6:]]:
    p.set_cov_splits(ns)
    p.iterate(1); p.sync()
    ts = [p.t_time_stage('weighted_cov', 20) * 1e3 for _ in range(3)]
    tu = p.t_time_stage('ip_update', 20) * 1e3
    print(f"cov splits {p.cov_splits():3d}: cov {min(ts):6.1f} us ({' '.join(f'{t:.1f}' for t in ts)}), update {tu:5.1f} us", flush=True)
