#!/usr/bin/env python3
"""GPU box: per-stage kernel times and the graph-replayed four-launch iteration at the shapes of the reference's own sweep
(overiva_sim_config.json: 2..8 microphones, 1..4 targets and determined AuxIVA; 4096-point STFT -> 2049 bins x ~235 frames),
in the three arithmetic modes.  -> profiles/rNN_stage_times_reference_shapes.log"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import overiva_amd as oa

T, F = 235, 2049
SHAPES = [(2, 1), (2, 2), (3, 1), (3, 2), (3, 3), (4, 2), (4, 3), (4, 4), (5, 2), (5, 3), (5, 5), (6, 2), (6, 3), (6, 4), (6, 6),
          (7, 1), (7, 2), (7, 3), (7, 7), (8, 2), (8, 3), (8, 4), (8, 8)]
g = torch.Generator(device="cuda"); g.manual_seed(1)
for M, K in SHAPES:
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    for mode in ("precise", "mixed", "fast"):
        p = oa.Plan(T, F, M, K, "laplace")
        p.set_precision(mode)
        p.set_x_device(X.data_ptr(), X)
        p.covariance(); p.set_w(None); p.iterate(2); p.sync()
        st = {s: round(min(p.t_time_stage(s, 10) for _ in range(3)) * 1e3, 1) for s in ("demix_power", "activation", "weighted_cov", "ip_update")}
        p.use_graph(True); p.iterate(8); p.sync()
        n = 100
        dt = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); p.iterate(n); p.sync(); dt = min(dt, time.perf_counter() - t0)
        res = ""
        if p.resident_info()["qualifies"] and (mode != "precise" or M == 4):
            p.use_graph(False); p.set_resident(True); p.iterate(20); p.sync()
            dr = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); p.iterate(100); p.sync(); dr = min(dr, time.perf_counter() - t0)
            res = f" | X-resident {dr / 100 * 1e6:.1f} us"
        print(f"({T}, {F}, {M}, {K}) {mode} {st} {n / dt:.0f} it/s ({dt / n * 1e6:.1f} us) splits {p.cov_splits()}{res}", flush=True)
        p.close()
