#!/usr/bin/env python3
"""GPU box: the demix + power stage at configs[4] (2048 x 4000 x 16 / 16), event-bracketed; OIVA_LIB selects a variant build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
T, F, M, K = 4000, 2048, 16, 16
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.sync()
ts = [p.t_time_stage("demix_power", 20) * 1e3 for _ in range(5)]
print(os.environ.get("OIVA_LIB", "default"), "demix_power us:", " ".join(f"{t:.1f}" for t in ts))
