#!/usr/bin/env python3
"""GPU box: configs[4] (2048 x 4000 x 16 / 16, mixed) stage times with the per-bin update of this process's $OIVA_DET16_ROWS
(1: one matrix row per lane, kernels_update16r.hip; 0: one matrix per wave, kernels_update16.hip), and W after 3 iterations
against the other form when its result is found in gpurun_out/det16_w_<other>.npy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (4000, 2048, 16, 16)
tag = os.environ.get("OIVA_DET16_ROWS", "1") + os.environ.get("OIVA_HMFMA_INNER", "")
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(3); p.sync()
W = p.get_w(np.complex128)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/det16_w_{tag}_{F}x{T}x{M}.npy", W)
import hashlib
print(f"rows={tag}: sha256(W after 3 iterations)[:16] = {hashlib.sha256(W.tobytes()).hexdigest()[:16]}")
import glob
for other in sorted(glob.glob(f"gpurun_out/det16_w_*_{F}x{T}x{M}.npy")):
    if other.endswith(f"det16_w_{tag}_{F}x{T}x{M}.npy"): continue
    Wo = np.load(other)
    print(f"rows={tag}: W vs {os.path.basename(other)} after 3 iterations: rel {np.linalg.norm(W - Wo) / np.linalg.norm(Wo):.3e}, bit-equal {np.array_equal(W, Wo)}, finite {np.isfinite(W).all()}")
for st in ("demix_power", "activation", "weighted_cov", "ip_update"):
    t = min(p.t_time_stage(st, 10) * 1e3 for _ in range(3))
    print(f"rows={tag} {F}x{T}x{M}: {st:14s} {t:8.1f} us", flush=True)
import time
p.use_graph(True); p.iterate(96); p.sync()
t0 = time.perf_counter(); p.iterate(96); p.sync(); dt = time.perf_counter() - t0
print(f"rows={tag}: {dt / 96 * 1e6:.1f} us per iteration = {96 / dt:.1f} it/s")
