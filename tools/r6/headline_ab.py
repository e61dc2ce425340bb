#!/usr/bin/env python3
"""GPU box: the headline shape (2048 x 4000 x 8 / 2, mixed) with this process's library ($OIVA_LIB): covariance and update stages, and
microseconds per iteration over three timings of 400 graph replays; run alternately with two libraries for an A/B on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
T, F, M, K = 4000, 2048, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(3); p.sync()
tc = min(p.t_time_stage("weighted_cov", 20) * 1e3 for _ in range(5))
tu = min(p.t_time_stage("ip_update", 20) * 1e3 for _ in range(5))
p.use_graph(True); p.iterate(400); p.sync()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); p.iterate(400); p.sync(); ts.append((time.perf_counter() - t0) / 400 * 1e6)
print(f"{os.path.basename(os.environ.get('OIVA_LIB', 'liboveriva_hip.so')):28s} cov {tc:6.1f}  update {tu:5.1f}  iteration " + " ".join(f"{t:.1f}" for t in ts) + " us", flush=True)
