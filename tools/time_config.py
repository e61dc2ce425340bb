#!/usr/bin/env python3
"""GPU box: per-stage kernel times and iterations/s for one shape:  T F M K [model]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]]
model = sys.argv[5] if len(sys.argv) > 5 else "laplace"
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
torch.cuda.synchronize()
p = oa.Plan(T, F, M, K, model)
p.set_x_device(X.data_ptr(), X)
p.covariance(); p.set_w(None); p.iterate(2); p.sync()
st = {s: p.t_time_stage(s, 5) * 1e3 for s in ("demix_power", "activation", "weighted_cov", "ip_update")}
res = {}
for graph in (0, 1):
    p.use_graph(bool(graph)); p.iterate(3); p.sync()
    n = 30
    t0 = time.perf_counter(); p.iterate(n); p.sync(); dt = time.perf_counter() - t0
    res[graph] = n / dt
import numpy as np
ok = bool(np.all(np.isfinite(p.get_w())))
print(f"T{T} F{F} M{M} K{K} {model}: stages us { {k: round(v, 1) for k, v in st.items()} }  sum {sum(st.values()):.0f} us | "
      f"it/s eager {res[0]:.0f} graph {res[1]:.0f} | cov splits {p.cov_splits()} finite {ok}")
