#!/usr/bin/env python3
"""GPU box: run one stage of the iteration a few times (for counter collection):  T F M K stage [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
from overiva_amd import _lib

T, F, M, K = [int(a) for a in sys.argv[1:5]]
stage = sys.argv[5]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace")
p.set_precision(_lib.PREC_FAST)
p.set_x_device(X.data_ptr(), X)
p.covariance(); p.set_w(None); p.iterate(1); p.sync()
print(stage, p.t_time_stage(stage, n) * 1e3, "us")
p.close()
