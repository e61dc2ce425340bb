#!/usr/bin/env python3
"""GPU box: microseconds per iteration of the four-launch path replayed from graphs, one stream against the two-branch form
(oiva_plan_set_split), at the headline shape and at its 2- and 4-GPU shards.  [bins ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, M, K = 4000, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
for F in [int(a) for a in sys.argv[1:]] or [2048, 1024, 512]:
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    for split in (0, 1, 0, 1):
        p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
        p.use_graph(True); p.set_split(split); p.iterate(32); p.sync()
        ts = []
        for r in range(9):
            t0 = time.perf_counter(); p.iterate(64); p.sync(); ts.append(time.perf_counter() - t0)
        print(f"{F} bins split {split}: {sorted(ts)[len(ts) // 2] / 64 * 1e6:7.2f} us/iter (min {min(ts) / 64 * 1e6:.2f})", flush=True)
        p.close()
