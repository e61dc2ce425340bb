#!/usr/bin/env python3
"""GPU box: does the 2049th bin (one workgroup more than two per CU) cost the per-bin update a round?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
g = torch.Generator(device="cuda"); g.manual_seed(1)
T = 235
for M, K in ((8, 4), (8, 2), (5, 5)):
    for F in (2040, 2048, 2049, 2052, 2304, 3072):
        X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
        p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.set_resident(False); p.iterate(3); p.sync()
        ts = sorted(p.t_time_stage("ip_update", 20) * 1e3 for _ in range(5))
        tp = sorted(p.t_time_stage("demix_power", 20) * 1e3 for _ in range(5))
        tc = sorted(p.t_time_stage("weighted_cov", 20) * 1e3 for _ in range(5))
        print(f"({T}, {F}, {M}, {K}): update {ts[0]:.2f} us  power {tp[0]:.2f}  cov {tc[0]:.2f} (splits {p.cov_splits()})", flush=True)
        p.close()
