#!/usr/bin/env python3
"""GPU box: one rank's shard of the headline shape (256 x 4000 x 8 / 2, mixed) in the X-resident kernel -- microseconds per
iteration at several launch lengths, single rank and loop-back world 8, with workgroup 0's phases.  [T F M K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (4000, 256, 8, 2)
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
for n in (20, 50, 200):
    for lb in (0, 8):
        p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
        if lb:
            p.resident_loopback(lb)
        p.set_resident(True); p.iterate(10); p.sync()
        ts = []
        for r in range(7):
            t0 = time.perf_counter(); p.iterate(n); p.sync(); ts.append(time.perf_counter() - t0)
        ph, _ = p.resident_phases()
        print(f"loopback {lb} steps {n:3d}: {sorted(ts)[len(ts) // 2] / n * 1e6:6.2f} us/iter (min {min(ts) / n * 1e6:.2f})", {k: round(v, 2) for k, v in ph.items()},
              "fallbacks", p.resident_info()["fallbacks"])
        p.close()
