#!/usr/bin/env python3
"""GPU box: per-stage kernel times per arithmetic mode:  T F M K [model [cov_splits]]
modes: fast (fp32), upd64 (fp32 covariance + float64 per-bin algebra), precise (float64 covariance + algebra)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import overiva_amd as oa
from overiva_amd import _lib

T, F, M, K = [int(a) for a in sys.argv[1:5]]
model = sys.argv[5] if len(sys.argv) > 5 else "laplace"
splits = int(sys.argv[6]) if len(sys.argv) > 6 else 0
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
torch.cuda.synchronize()
for name, flags in (("fast", _lib.PREC_FAST), ("upd64", _lib.PREC_UPDATE_F64), ("precise", _lib.PREC_PRECISE)):
    p = oa.Plan(T, F, M, K, model)
    p.set_precision(flags)
    if splits:
        p.set_cov_splits(splits)
    p.set_x_device(X.data_ptr(), X)
    p.covariance(); p.set_w(None); p.iterate(2); p.sync()
    st = {s: p.t_time_stage(s, 10) * 1e3 for s in ("demix_power", "activation", "weighted_cov", "ip_update")}
    p.use_graph(True); p.iterate(8); p.sync()
    n = 48
    t0 = time.perf_counter(); p.iterate(n); p.sync(); dt = time.perf_counter() - t0
    ok = bool(np.all(np.isfinite(p.get_w())))
    print(f"T{T} F{F} M{M} K{K} {model} {name:8s}: us { {k: round(v, 1) for k, v in st.items()} } sum {sum(st.values()):.0f} | "
          f"graph {n / dt:.0f} it/s ({dt / n * 1e6:.0f} us) | cov splits {p.cov_splits()} finite {ok}", flush=True)
    p.close()
