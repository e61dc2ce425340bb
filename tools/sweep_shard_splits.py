#!/usr/bin/env python3
"""GPU box: covariance / update / power times of a bin shard against the number of frame splits:  T F M K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, overiva_amd as oa
T, F, M, K = [int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (4000, 256, 8, 2))]
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
for ns in (0, 8, 16, 24, 31, 48, 62):
    p.set_cov_splits(ns)
    p.iterate(1); p.sync()
    print(f"cov splits {p.cov_splits():3d}: cov {p.t_time_stage('weighted_cov', 20) * 1e3:6.1f} us, update {p.t_time_stage('ip_update', 20) * 1e3:6.1f} us", flush=True)
p.set_cov_splits(0)
for ns in (0, 16, 31, 62, 125):
    p.set_pow_splits(ns)
    print(f"pow splits {ns:3d}: power {p.t_time_stage('demix_power', 20) * 1e3:6.1f} us", flush=True)
