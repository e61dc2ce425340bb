#!/usr/bin/env python3
"""GPU box: the per-bin update stage (event-bracketed) at a few shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
g = torch.Generator(device="cuda"); g.manual_seed(1)
for T, F, M, K in ((4000, 2048, 8, 2), (235, 2049, 8, 4), (235, 2049, 7, 3), (235, 2049, 5, 5), (235, 2049, 8, 2), (4000, 1024, 8, 2), (4000, 2048, 16, 16)):
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(3); p.sync()
    ts = sorted(p.t_time_stage("ip_update", 20) * 1e3 for _ in range(5))
    p.use_graph(True); p.iterate(60); p.sync()
    import time
    dt = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); p.iterate(100); p.sync(); dt = min(dt, time.perf_counter() - t0)
    print(f"({T}, {F}, {M}, {K}) splits {p.cov_splits()}: update {ts[0]:.2f} (median {ts[2]:.2f}) us; iteration {dt / 100 * 1e6:.1f} us", flush=True)
    p.close()
