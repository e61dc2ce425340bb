#!/usr/bin/env python3
"""GPU box: is the 16-channel demix + power pass data dependent?  Same launch, W dense vs W zero vs X zero."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import overiva_amd as oa
from overiva_amd import _lib

T, F, M, K = 4000, 2048, 16, 16
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
Z = torch.zeros_like(X)
rng = np.random.default_rng(0)
Wd = (rng.standard_normal((F, M, K)) + 1j * rng.standard_normal((F, M, K))).astype(np.complex64)
for name, x, w in (("dense X, dense W", X, Wd), ("dense X, zero W", X, np.zeros_like(Wd)), ("zero X, dense W", Z, Wd)):
    p = oa.Plan(T, F, M, K, "laplace")
    p.set_precision(_lib.PREC_FAST)
    p.set_x_device(x.data_ptr(), x)
    p.covariance()
    p.set_w(w)
    us = p.t_time_stage("demix_power", 20) * 1e3
    print(f"{name}: demix_power {us:.1f} us", flush=True)
    p.close()
