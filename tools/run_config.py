#!/usr/bin/env python3
"""GPU box: run N iterations of one shape (for rocprofv3):  T F M K [n_iter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
torch.cuda.synchronize()
p = oa.Plan(T, F, M, K, "laplace")
p.set_x_device(X.data_ptr(), X)
p.covariance(); p.set_w(None); p.iterate(n); p.sync()
print("done", T, F, M, K, n)
