#!/usr/bin/env python3
"""GPU box: does launching another graph (iterate(5)) before the timed iterate(20) make that one slower?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
T, F, M, K = 4000, 2048, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
def tm(p, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); p.iterate(n); p.sync(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for rep in range(3):
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.use_graph(True)
    seq = [20, 20, 20, 5, 20, 20, 20, 5, 5, 20, 20, 1, 20, 20]
    print(" ".join(f"{n}:{tm(p, n):.1f}" for n in seq), flush=True)
    p.close()
