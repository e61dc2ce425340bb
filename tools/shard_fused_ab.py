#!/usr/bin/env python3
"""GPU box: one rank's shard of the headline shape at 2 / 4 GPUs (1024 / 512 bins x 4000 x 8 / 2, mixed), the four-launch
iteration replayed from a graph WITHOUT and WITH the in-kernel exchange (loop-back), alternating in one process: wall time per
iteration and the event-bracketed stages -- where the exchange's microseconds go."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, M, K = 4000, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
for F, world in ((1024, 2), (512, 4)):
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    for rep in range(3):
        for lb in (0, world):
            p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
            if lb:
                p.fused_loopback(lb)
            p.iterate(2); p.sync()
            st = {s: round(min(p.t_time_stage(s, 20) for _ in range(3)) * 1e3, 2) for s in ("demix_power", "activation", "weighted_cov", "ip_update")}
            p.use_graph(True); p.iterate(20); p.sync()
            ts = []
            for r in range(7):
                t0 = time.perf_counter(); p.iterate(100); p.sync(); ts.append(time.perf_counter() - t0)
            print(f"F {F} loopback {lb}: {sorted(ts)[3] / 100 * 1e6:7.2f} us/iter (min {min(ts) / 100 * 1e6:.2f})  stages {st}  splits {p.cov_splits()}", flush=True)
            p.close()
