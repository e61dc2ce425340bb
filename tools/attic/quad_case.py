#!/usr/bin/env python3
"""GPU box: n eager iterations of a 9..16-channel plan on an iid tensor (for rocprofv3 around it):  T F M K [n [mode [quad]]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
mode = sys.argv[6] if len(sys.argv) > 6 else "fast"
quad = int(sys.argv[7]) if len(sys.argv) > 7 else 1
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision(mode); p.set_cov_quad(bool(quad)); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
p.iterate(n)
p.sync()
tot, per = p.iterate_timed(n, per_kernel=True)
print({k: round(v / n * 1e3, 1) for k, v in per.items()})
