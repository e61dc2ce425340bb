#!/usr/bin/env python3
"""GPU box, with the -DOIVA_COVDMA_TRACE variant of the library (tools/build_variant.py cdtrace "-DOIVA_COVDMA_TRACE" kernels_cov.hip;
OIVA_LIB=overiva_amd/liboveriva_hip_cdtrace.so): 100 MHz wall-clock stamps of every workgroup of cov_dma_kernel<8, 2> -- start, prologue done
(sum over the frames), loop done, DMA queue drained, epilogue done -- at the headline shape and at a 4-GPU shard's 512 bins."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, overiva_amd as oa
from overiva_amd import _lib
lib = _lib.load()
lib.oiva_debug_covdma_trace.argtypes = [C.c_void_p]
for F in (2048, 512):
    T, M, K = 4000, 8, 2
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.set_resident(False); p.iterate(3); p.sync()
    tc = min(p.t_time_stage("weighted_cov", 20) * 1e3 for _ in range(5))
    p.iterate(1); p.sync()                   # the trace of a launch inside an iteration (after the activation kernel)
    nwg = (F // 16) * p.cov_splits()
    out = np.zeros(8 * 2048, np.int64)
    lib.oiva_debug_covdma_trace(out.ctypes.data)
    t = out.reshape(2048, 8)[:nwg, :5].astype(np.float64) / 100.0          # us
    t -= t[:, 0].min()
    d = np.diff(t, axis=1)
    q = lambda a: f"min {a.min():6.2f}  median {np.median(a):6.2f}  max {a.max():6.2f}"
    print(f"{F} bins, {p.cov_splits()} frame splits, {nwg} workgroups; stage timer {tc:.1f} us")
    print(f"  start             {q(t[:, 0])}")
    print(f"  prologue          {q(d[:, 0])}")
    print(f"  loop              {q(d[:, 1])}")
    print(f"  drain             {q(d[:, 2])}")
    print(f"  epilogue          {q(d[:, 3])}")
    print(f"  loop done at      {q(t[:, 2])}")
    print(f"  workgroup done at {q(t[:, 4])}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    np.save(f"gpurun_out/covdma_trace_{F}.npy", t)
    nx = F // 16
    tt = t.reshape(p.cov_splits(), nx, 5)                    # [split y][bin group x]
    print("  loop time by frame split y:          " + " ".join(f"{np.median(tt[y, :, 2] - tt[y, :, 1]):6.1f}" for y in range(tt.shape[0])))
    print("  loop time by x mod 8 (XCD if round-robin): " + " ".join(f"{np.median((tt[:, k::8, 2] - tt[:, k::8, 1])):6.1f}" for k in range(8)))
    print("  loop time by x // 16:                " + " ".join(f"{np.median((tt[:, 16 * k:16 * k + 16, 2] - tt[:, 16 * k:16 * k + 16, 1])):6.1f}" for k in range(nx // 16)), flush=True)
    p.close()
