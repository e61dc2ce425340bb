#!/usr/bin/env python3
"""GPU box: the power pass of shapes power_lds_kernel does not take (a bin count that is no multiple of 64, fewer than 16 channels) with this process's
library ($OIVA_LIB) -- power_mfma_kernel; run alternately with two libraries.  (The stage timer launches
the kernel twenty times back to back, where the order of the bins made no difference for power_lds_kernel either; the iteration is what counts.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
g = torch.Generator(device="cuda"); g.manual_seed(1)
for (T, F, M, K) in ((4000, 2049, 16, 16), (4000, 2048, 12, 12), (4000, 2049, 14, 9), (235, 2049, 16, 16)):
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
    tp = sorted(p.t_time_stage("demix_power", 10) * 1e3 for _ in range(5))
    n = 60 if T > 1000 else 300
    p.use_graph(True); p.iterate(n); p.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); p.iterate(n); p.sync(); ts.append((time.perf_counter() - t0) / n * 1e6)
    print(f"{os.path.basename(os.environ.get('OIVA_LIB', 'liboveriva_hip.so')):26s} ({T}, {F}, {M}, {K}): demix_power back to back {tp[0]:7.1f} us; iteration " + " ".join(f"{t:.1f}" for t in ts), flush=True)
    p.close(); del X
