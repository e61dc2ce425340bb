#!/usr/bin/env python3
"""GPU box: the headline shape (2048 x 4000 x 8 / 2, mixed) against forced numbers of frame splits of the covariance pass (4 = one workgroup
round of 512; more = later workgroups start when the first finish)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (4000, 2048, 8, 2)
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
for ns in (0, 4, 5, 6, 7, 8, 10, 12, 16, 0):
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed")
    if ns:
        p.set_cov_splits(ns)
    p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.set_resident(False); p.iterate(3); p.sync()
    tc = min(p.t_time_stage("weighted_cov", 20) * 1e3 for _ in range(5))
    tu = min(p.t_time_stage("ip_update", 20) * 1e3 for _ in range(5))
    p.use_graph(True); p.iterate(300); p.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); p.iterate(300); p.sync(); ts.append((time.perf_counter() - t0) / 300 * 1e6)
    print(f"splits {ns:2d} -> {p.cov_splits():2d}: cov {tc:6.1f}  update {tu:5.1f}  iteration " + " ".join(f"{t:.1f}" for t in ts) + " us", flush=True)
    p.close()
