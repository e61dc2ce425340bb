#!/usr/bin/env python3
"""GPU box: the headline's covariance pass (cov_dma_kernel<8, 2>, 2048 x 4000 x 8 / 2) with parts of it compiled out (variant builds
`tools/build_variant.py cdaN "-DOIVA_COVDMA_ABLATE=N" kernels_cov.hip`: 1 no arithmetic, 2 no epilogue, 4 no sum over the frames in the
prologue, 7 all three = the ring alone) -- where the 11 us between the kernel and its stream (tools/r6/dmabench.hip row Q) go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, overiva_amd as oa
T, F, M, K = 4000, 2048, 8, 2
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(3); p.sync()
ts = sorted(p.t_time_stage("weighted_cov", 20) * 1e3 for _ in range(5))
tp = sorted(p.t_time_stage("demix_power", 20) * 1e3 for _ in range(5))
print(f"{os.path.basename(os.environ.get('OIVA_LIB', 'liboveriva_hip.so')):28s} weighted_cov {ts[0]:6.1f} us (median {ts[2]:.1f})   demix_power {tp[0]:6.1f}", flush=True)
