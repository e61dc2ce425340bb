#!/usr/bin/env python3
"""GPU box: the update stage of configs[4] with the library of $OIVA_LIB (variant builds with one role of update_det16r_kernel
alone: timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, overiva_amd as oa
T, F, M, K = 4000, 2048, 16, 16
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
p.t_time_stage("demix_power", 1); p.t_time_stage("activation", 1); p.t_time_stage("weighted_cov", 1)
print(os.environ.get("OIVA_LIB", "default"), "ip_update", min(p.t_time_stage("ip_update", 10) * 1e3 for _ in range(3)), "us")
