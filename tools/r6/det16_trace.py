#!/usr/bin/env python3
"""GPU box, with the -DOIVA_R16_TRACE variant of the library (tools/build_variant.py r16trace "-DOIVA_R16_TRACE" kernels_update16r.hip;
OIVA_LIB=overiva_amd/liboveriva_hip_r16trace.so): clock stamps (100 MHz) of workgroup 0 of update_det16r_kernel at configs[4]."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, overiva_amd as oa
from overiva_amd import _lib
T, F, M, K = 4000, 2048, 16, 16
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None); p.iterate(3); p.sync()
print("ip_update", min(p.t_time_stage("ip_update", 10) * 1e3 for _ in range(3)), "us")
lib = _lib.load()
out = np.zeros(3 * 80, np.uint64)
lib.oiva_debug_r16_trace.argtypes = [C.c_void_p]
rc = lib.oiva_debug_r16_trace(out.ctypes.data)
t = out.reshape(3, 80).astype(np.int64)
t0 = t[0, 64]
ghz = float(os.environ.get("CLK_GHZ", "2.1"))
us = lambda v: (v - t0) / (ghz * 1e3)        # (s_memtime runs at the shader clock here)
print(f"wave 0: start 0.00, C-phase done {us(t[0, 65]):.2f} us")
print("C-phase step 8: start, argmax done, lds quiet, row written, row read, (step 9 start):", [round(us(v), 3) for v in (t[0, 66], t[0, 67], t[0, 68], t[0, 69], t[0, 70], t[0, 72])])
for s_ in range(16):
    a = t[0, 4 * s_: 4 * s_ + 4]
    print(f"src {s_:2d}: A top {us(a[0]):6.2f} wait-from {us(a[1]):6.2f} got {us(a[2]):6.2f} end {us(a[3]):6.2f}")
for w in (1, 2):
    rows = [(us(t[w, 4 * s_]), s_) for s_ in range(16) if t[w, 4 * s_] > 0 and abs(us(t[w, 4 * s_])) < 1e4]
    for _, s_ in sorted(rows):
        b = t[w, 4 * s_: 4 * s_ + 4]
        print(f"wave {w} src {s_:2d}: top {us(b[0]):6.2f} partials {us(b[1]):6.2f} solved {us(b[2]):6.2f} handed {us(b[3]):6.2f}")

hw = np.zeros(2 * 3 * 1024, np.uint32)
lib.oiva_debug_r16_hwid.argtypes = [C.c_void_p]
lib.oiva_debug_r16_hwid(hw.ctypes.data)
hw = hw.reshape(1024, 3, 2)[:512]
simd = (hw[:, :, 0] >> 4) & 3; cu = (hw[:, :, 0] >> 8) & 15; se = (hw[:, :, 0] >> 13) & 7; sh = (hw[:, :, 0] >> 12) & 1; xcc = hw[:, :, 1] & 15
print("first workgroups: (xcc, se, sh, cu, simd) per wave A, B0, B1")
for wg in range(12):
    print(wg, [(int(xcc[wg, w]), int(se[wg, w]), int(sh[wg, w]), int(cu[wg, w]), int(simd[wg, w])) for w in range(3)])
# per physical SIMD: which roles landed there
from collections import Counter
load = {}
for wg in range(512):
    for w in range(3):
        load.setdefault((int(xcc[wg, w]), int(se[wg, w]), int(sh[wg, w]), int(cu[wg, w]), int(simd[wg, w])), []).append("ABB"[w])
pat = Counter("".join(sorted(v)) for v in load.values())
print("role sets per occupied SIMD:", dict(pat), " occupied SIMDs:", len(load))
