#!/usr/bin/env python3
"""Tuning helper (GPU box): per-stage kernel times on the headline workload, with ablation masks."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa

T, F, M, K = [int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (4000, 2048, 8, 2))]
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
torch.cuda.synchronize()
p = oa.Plan(T, F, M, K, "laplace")
p.set_x_device(X.data_ptr(), X)
p.covariance(); p.set_w(None); p.iterate(3); p.sync()
for st in ("demix_power", "activation", "weighted_cov", "ip_update"):
    print(f"{st:14s} {p.t_time_stage(st, 20) * 1e3:8.1f} us")
for mask, what in ((1, "loads+stores only"), (2, "1 pivot instead of 8 in the IP solve"), (4, "no J update"), (6, "1 pivot, no J")):
    p.t_set_flags(mask << 8)
    print(f"ip_update dbg={mask} ({what}): {p.t_time_stage('ip_update', 20) * 1e3:8.1f} us")
p.t_set_flags(0)
p.t_set_flags(2); print(f"ip_update row layout: {p.t_time_stage('ip_update', 20) * 1e3:8.1f} us")
p.t_set_flags(1); print(f"ip_update fp64: {p.t_time_stage('ip_update', 20) * 1e3:8.1f} us")
p.t_set_flags(0)
for ns in (16, 24, 32, 48, 64, 96, 128):
    p.set_pow_splits(ns)
    print(f"pow splits {ns:3d}: {p.t_time_stage('demix_power', 20) * 1e3:8.1f} us")
p.set_pow_splits(0)
for ns in (4, 8, 12, 16):
    p.set_cov_splits(ns)
    print(f"cov splits {p.cov_splits():3d}: {p.t_time_stage('weighted_cov', 20) * 1e3:8.1f} us")
