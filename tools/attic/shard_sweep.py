#!/usr/bin/env python3
"""GPU box: the four-launch iteration of the 2- and 4-GPU shards (1024 / 512 bins x 4000 x 8 / 2, mixed) against the frame
splits of the power and covariance passes: stage times and the graph-replayed iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, overiva_amd as oa
T, M, K = 4000, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
for F in (512, 1024):
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
    print(f"== {F} bins: default cov splits {p.cov_splits()}")
    for ns in (4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32):
        p.set_cov_splits(ns); p.iterate(1); p.sync()
        tc = min(p.t_time_stage('weighted_cov', 30) * 1e3 for _ in range(3)); tu = min(p.t_time_stage('ip_update', 30) * 1e3 for _ in range(3))
        print(f"  cov splits {p.cov_splits():3d}: cov {tc:6.1f} us, update {tu:5.1f} us, sum {tc + tu:6.1f}", flush=True)
    p.set_cov_splits(0)
    for ns in (4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
        p.set_pow_splits(ns); p.iterate(1); p.sync()
        tp = min(p.t_time_stage('demix_power', 30) * 1e3 for _ in range(3)); ta = min(p.t_time_stage('activation', 30) * 1e3 for _ in range(3))
        print(f"  pow splits {ns:3d}: power {tp:6.1f} us, activation {ta:5.1f} us", flush=True)
    p.close()
