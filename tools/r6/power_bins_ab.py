#!/usr/bin/env python3
"""GPU box: 16 x 16 at 4000 frames against the number of bins -- whole batches on a power-of-two row stride (2048), whole batches on another stride (2112,
1984), a ragged last batch (2049, 2111); $OIVA_POWER_LDS=0 in the environment: the frame-major kernel instead of power_lds_kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
g = torch.Generator(device="cuda"); g.manual_seed(1)
T, M, K = 4000, 16, 16
for F in (2048, 2049, 2112, 2111, 1984):
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(2); p.sync()
    st = {s: min(p.t_time_stage(s, 10) * 1e3 for _ in range(3)) for s in ("demix_power", "weighted_cov", "ip_update")}
    p.use_graph(True); p.iterate(60); p.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); p.iterate(60); p.sync(); ts.append((time.perf_counter() - t0) / 60 * 1e6)
    print(f"POWER_LDS={os.environ.get('OIVA_POWER_LDS', '1')} {F} bins: stages " + " ".join(f"{k} {v:.1f}" for k, v in st.items()) + "; iteration " + " ".join(f"{t:.1f}" for t in ts) + f"  ({min(ts) / F * 2048:.1f} per 2048 bins)", flush=True)
    p.close(); del X
