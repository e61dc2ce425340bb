#!/usr/bin/env python3
"""GPU box: n iterations of the X-resident kernel on an iid tensor (for rocprofv3 around it):  T F M K [n [mode]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 50
mode = sys.argv[6] if len(sys.argv) > 6 else "mixed"
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision(mode); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
p.set_resident(True)
for _ in range(3):
    p.iterate(n)
p.sync()
print(p.resident_info(), p.resident_phases())
