"""GPU box: end-to-end time of the drop-in call with host arrays at the shapes the reference's own drivers produce
(overiva_oneshot.py: STFT 4096 -> 2049 bins, ~160 frames, 4 mics, complex128, 20-100 iterations), stage by stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import overiva_amd as oa
from oracle import overiva_oracle as orc

for (T, F, M, K) in ((160, 2049, 4, 2), (235, 2049, 8, 2), (235, 2049, 8, 4)):
    X64 = orc.synth_mixture(T, F, M, K, seed=0)
    for dt in (np.complex128, np.complex64):
        X = X64.astype(dt)
        for n_iter in (20, 100):
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                Y = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True)
                ts.append(time.perf_counter() - t0)
            info = oa.last_solver_info()
            print(f"{F}x{T}x{M}/{K} {np.dtype(dt).name} n_iter={n_iter}: {' '.join(f'{1e3 * t:.2f}' for t in ts)} ms  ({info['precision']}, resident launches {info['resident_launches']})", flush=True)
    # stage by stage, complex128
    X = X64.astype(np.complex128)
    for it in range(2):
        t0 = time.perf_counter()
        p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("precise"); t1 = time.perf_counter()
        p.set_x(X); t2 = time.perf_counter()
        p.covariance(); p.set_w(None); p.sync(); t3 = time.perf_counter()
        p.iterate(100); p.sync(); t4 = time.perf_counter()
        Y = p.demix(True, dtype=np.complex128); t5 = time.perf_counter()
        W = p.get_w(np.complex128); p.close(); t6 = time.perf_counter()
        print(f"  stages: create {1e3*(t1-t0):.2f} | upload X {1e3*(t2-t1):.2f} | prologue {1e3*(t3-t2):.2f} | 100 its {1e3*(t4-t3):.2f} | demix+download {1e3*(t5-t4):.2f} | W+close {1e3*(t6-t5):.2f} | total {1e3*(t6-t0):.2f} ms", flush=True)
