"""GPU box: end-to-end time of the drop-in call with host arrays at the headline shape ($OIVA_DEMIX_IO = legacy | ring | register:
the form of the final hand-over of Y, csrc/plan.hip demix_to_host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import overiva_amd as oa
from oracle import overiva_oracle as orc
T, F, M, K = 4000, 2048, 8, 2
print("OIVA_DEMIX_IO =", os.environ.get("OIVA_DEMIX_IO", "(default)"), " OIVA_IO_THREADS =", os.environ.get("OIVA_IO_THREADS", "(default)"))
X = orc.synth_iid(T, F, M, seed=0)
Y = None
for it in range(3):
    del Y           # (unmapping the previous 131 MB result is not part of any stage below)
    t0 = time.perf_counter()
    p = oa.Plan(T, F, M, K, "laplace"); t1 = time.perf_counter()
    p.set_x(X); t2 = time.perf_counter()
    p.covariance(); p.set_w(None); p.use_graph(True); p.sync(); t3 = time.perf_counter()
    p.iterate(20); p.sync(); t4 = time.perf_counter()
    Y = p.demix(True); t5 = time.perf_counter()
    W = p.get_w(); p.close(); t6 = time.perf_counter()
    print(f"run {it}: create {1e3*(t1-t0):.1f} ms | upload X {1e3*(t2-t1):.1f} ms ({X.nbytes/1e9/(t2-t1):.1f} GB/s) | prologue {1e3*(t3-t2):.1f} | "
          f"20 its {1e3*(t4-t3):.1f} | demix+download Y {1e3*(t5-t4):.1f} ms ({Y.nbytes/1e9/(t5-t4):.1f} GB/s) | W+close {1e3*(t6-t5):.1f} | total {1e3*(t6-t0):.1f} ms")
del Y
# (the result of the previous call is dropped OUTSIDE the timed region: unmapping 131 MB is the caller's 3-5 ms, whoever made the array)
for _ in range(4):
    t0 = time.perf_counter(); Y = oa.overiva(X, n_src=K, n_iter=20); t1 = time.perf_counter(); print(f"overiva() 20 its end to end: {1e3*(t1-t0):.1f} ms"); del Y
for dt in (np.complex64, np.complex128):
    Xd = X.astype(dt)
    for mode in ("precise", "fast"):
        oa.set_precision(mode)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); Y = oa.overiva(Xd, n_src=K, n_iter=20); ts.append(time.perf_counter() - t0); del Y
        print(f"overiva({np.dtype(dt).name}, {mode}) 20 its end to end: {' '.join(f'{1e3*t:.1f}' for t in ts)} ms")
