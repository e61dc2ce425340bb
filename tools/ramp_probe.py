#!/usr/bin/env python3
"""GPU box: does the first timed run after plan set-up pay a one-off cost, and does it depend on whether the warm-up
already replayed the batch graph?  For warm-up lengths W: wall time of iterate(20) + sync, first and second call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, overiva_amd as oa
from overiva_amd import _lib
T, F, M, K = 4000, 2048, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
for trial in range(3):
    for W in (5, 8, 40):
        p = oa.Plan(T, F, M, K, "laplace"); p.set_precision(_lib.PREC_FAST); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
        p.use_graph(True); p.iterate(W); p.sync(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); p.iterate(20); p.sync(); ts.append((time.perf_counter() - t0) * 1e6)
        print(f"warm-up {W:3d}: iterate(20) + sync = {ts[0]:7.1f}  {ts[1]:7.1f}  {ts[2]:7.1f} us", flush=True)
        p.close()
