#!/usr/bin/env python3
"""GPU box: configs[4] (2048 x 4000 x 16 / 16, mixed) -- microseconds per iteration of this process's library ($OIVA_LIB), three
timings of 96 graph replays; run alternately with two libraries for an A/B on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, overiva_amd as oa
T, F, M, K = 4000, 2048, 16, 16
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
p.use_graph(True); p.iterate(96); p.sync()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); p.iterate(96); p.sync(); ts.append((time.perf_counter() - t0) / 96 * 1e6)
print(os.path.basename(os.environ.get("OIVA_LIB", "liboveriva_hip.so")), " ".join(f"{t:.1f}" for t in ts), "us per iteration", flush=True)
