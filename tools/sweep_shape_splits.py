#!/usr/bin/env python3
"""GPU box: the covariance pass (and the update behind it, which adds the splits' partials) at the reference's sweep shapes
(2049 bins x 235 frames) for forced numbers of frame splits, `mixed`: is the geometry the plan picks the fastest one?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F = 235, 2049
g = torch.Generator(device="cuda"); g.manual_seed(1)
for M, K in ((7, 3), (8, 4), (5, 5), (6, 4), (8, 3), (7, 7), (8, 8), (6, 3), (5, 3), (7, 2)):
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    row = []
    for ns in (0, 1, 2, 3, 4, 6, 8):
        p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed")
        if ns:
            p.set_cov_splits(ns)
        p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
        c = min(p.t_time_stage("weighted_cov", 20) for _ in range(3)) * 1e3
        u = min(p.t_time_stage("ip_update", 20) for _ in range(3)) * 1e3
        p.use_graph(True); p.iterate(10); p.sync()
        dt = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); p.iterate(100); p.sync(); dt = min(dt, time.perf_counter() - t0)
        row.append(f"{ns}->{p.cov_splits()}: cov {c:.1f} upd {u:.1f} it {dt / 100 * 1e6:.1f}")
        p.close()
    print(f"({M}, {K})  " + " | ".join(row), flush=True)
