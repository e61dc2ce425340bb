#!/usr/bin/env python3
"""GPU box, under rocprofv3 --kernel-trace --stats: 200 graph-replayed iterations of ONE of the reference's sweep shapes
(2049 bins x 235 frames, M channels / K sources from the command line) in `mixed`, so that the per-kernel durations can be set
against the wall time of the iteration (what is launch overhead, what is kernel).  Prints the wall time per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa

M, K = int(sys.argv[1]), int(sys.argv[2])
T, F = 235, 2049
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace")
p.set_precision("mixed")
p.set_x_device(X.data_ptr(), X)
p.covariance(); p.set_w(None); p.use_graph(True); p.iterate(20); p.sync()
dt = 1e9
for _ in range(3):
    t0 = time.perf_counter(); p.iterate(60); p.sync(); dt = min(dt, time.perf_counter() - t0)
print(f"({T}, {F}, {M}, {K}) mixed: {dt / 60 * 1e6:.1f} us per iteration, splits {p.cov_splits()}")
p.close()
