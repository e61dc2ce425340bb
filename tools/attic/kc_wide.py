#!/usr/bin/env python3
"""GPU box: 7 channels / 3+ sources and 5 / 5 with $OIVA_COV_KC_WIDE (sources per pass of the generic covariance kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
g = torch.Generator(device="cuda"); g.manual_seed(1)
for T, F, M, K in ((235, 2049, 7, 3), (235, 2049, 5, 5), (235, 2049, 7, 7), (235, 2049, 7, 4), (4000, 2048, 7, 3), (4000, 2048, 5, 5), (4000, 2048, 7, 7), (1000, 513, 7, 3)):
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(3); p.sync()
    c = min(p.t_time_stage("weighted_cov", 10) for _ in range(3)) * 1e3
    p.use_graph(True); p.iterate(40); p.sync()
    dt = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); p.iterate(50); p.sync(); dt = min(dt, time.perf_counter() - t0)
    print(f"wide={os.environ.get('OIVA_COV_KC_WIDE', '1')} ({T}, {F}, {M}, {K}) splits {p.cov_splits()}: cov {c:.1f} us; iteration {dt / 50 * 1e6:.1f} us", flush=True)
    p.close()
