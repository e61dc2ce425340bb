#!/usr/bin/env python3
"""GPU box: fixed cost of one X-resident launch -- wall time of iterate(n) + sync for small n on the 8-GPU shard."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, overiva_amd as oa
T, F, M, K = 4000, 256, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
p.set_resident(True); p.iterate(10); p.sync()
for n in (1, 2, 5, 10, 20, 50, 100):
    ts = []
    for r in range(15):
        t0 = time.perf_counter(); p.iterate(n); p.sync(); ts.append(time.perf_counter() - t0)
    print(f"n {n:3d}: median {sorted(ts)[7] * 1e6:8.1f} us  min {min(ts) * 1e6:8.1f} us")
