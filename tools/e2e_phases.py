#!/usr/bin/env python3
"""GPU box: where the milliseconds of the drop-in call go at the headline shape -- the phases of overiva() one by one, with the
destination of Y fresh / pre-faulted, and the pre-fault alone (oiva_host_prefault) beside and without the upload of X."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import overiva_amd as oa
from overiva_amd import _lib
from oracle import overiva_oracle as orc
T, F, M, K = 4000, 2048, 8, 2
print("OIVA_DEMIX_IO =", os.environ.get("OIVA_DEMIX_IO", "(default)"), " OIVA_IO_THREADS =", os.environ.get("OIVA_IO_THREADS", "(default)"))
X = orc.synth_iid(T, F, M, seed=0)
try:
    print("transparent_hugepage:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| OIVA_IO_THP =", os.environ.get("OIVA_IO_THP", "(default on)"))
except OSError as e:
    print("transparent_hugepage: ?", e)
lib = _lib.load()
ms = lambda t: f"{1e3 * t:.2f}"
def prefault(a):
    lib.oiva_host_prefault(C.c_void_p(a.ctypes.data), a.nbytes)
for rep in range(3):
    a = np.empty((T, F, K), np.complex64); t0 = time.perf_counter(); prefault(a); t1 = time.perf_counter()
    b = np.empty((T, F, K), np.complex64); t2 = time.perf_counter(); b.fill(0); t3 = time.perf_counter()
    print(f"pre-fault of {a.nbytes >> 20} MB alone: {ms(t1 - t0)} ms ({a.nbytes / 1e9 / (t1 - t0):.1f} GB/s); numpy fill of a fresh array: {ms(t3 - t2)} ms")
    del a, b
for rep in range(3):
    for pre in ("none", "sync", "thread"):
        t = [time.perf_counter()]
        p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("mixed"); p.use_graph(True); t.append(time.perf_counter())
        out = np.empty((T, F, K), np.complex64); th = None
        if pre == "sync":
            prefault(out)
        elif pre == "thread":
            th = threading.Thread(target=prefault, args=(out,)); th.start()
        t.append(time.perf_counter())
        p.set_x(X); t.append(time.perf_counter())
        p.covariance(); p.set_w(None); p.iterate(20); p.sync(); t.append(time.perf_counter())
        if th: th.join()
        t.append(time.perf_counter())
        p.demix(True, out=out); t.append(time.perf_counter())
        W = p.get_w(np.complex128); t.append(time.perf_counter())
        p.close(); t.append(time.perf_counter())
        d = np.diff(t)
        print(f"pre-fault {pre:6s}: create {ms(d[0])} | alloc(+pre-fault) {ms(d[1])} | upload {ms(d[2])} | prologue + 20 its {ms(d[3])} | join {ms(d[4])} | demix + hand-over {ms(d[5])} "
              f"({out.nbytes / 1e9 / d[5]:.1f} GB/s) | get_w {ms(d[6])} | close {ms(d[7])} | total {ms(t[-1] - t[0])} ms", flush=True)
        del out
for rep in range(4):
    t0 = time.perf_counter(); Y = oa.overiva(X, n_src=K, n_iter=20); t1 = time.perf_counter()
    print(f"overiva() 20 its end to end: {ms(t1 - t0)} ms")
    del Y

# complex128 in and out (what the reference's own drivers pass, overiva_oneshot.py:293): phases
X128 = X.astype(np.complex128)
for rep in range(3):
    t = [time.perf_counter()]
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision("precise"); p.use_graph(True); t.append(time.perf_counter())
    p.set_x(X128); t.append(time.perf_counter())
    p.covariance(); p.set_w(None); p.iterate(20); t.append(time.perf_counter())
    out = np.empty((T, F, K), np.complex128); prefault(out); t.append(time.perf_counter())
    p.sync(); t.append(time.perf_counter())
    p.demix(True, out=out); t.append(time.perf_counter())
    W = p.get_w(np.complex128); p.close(); t.append(time.perf_counter())
    d = np.diff(t)
    print(f"complex128: create {ms(d[0])} | upload + conversion of {X128.nbytes >> 20} MB {ms(d[1])} ({X128.nbytes / 1e9 / d[1]:.1f} GB/s) | queue prologue + 20 its {ms(d[2])} | "
          f"alloc + pre-fault of {out.nbytes >> 20} MB {ms(d[3])} | wait for the iterations {ms(d[4])} | demix + hand-over {ms(d[5])} ({out.nbytes / 1e9 / d[5]:.1f} GB/s) | "
          f"get_w + close {ms(d[6])} | total {ms(t[-1] - t[0])} ms", flush=True)
    del out
for rep in range(3):
    t0 = time.perf_counter(); Y = oa.overiva(X128, n_src=K, n_iter=20); t1 = time.perf_counter()
    print(f"overiva(complex128) 20 its end to end: {ms(t1 - t0)} ms")
    del Y

# complex64 input in the `precise` arithmetic (not the default for this dtype): phases, then the drop-in call
for rep in range(4):
    t = [time.perf_counter()]
    p = oa.Plan(T, F, M, K, "laplace"); t.append(time.perf_counter())
    p.set_precision("precise"); t.append(time.perf_counter())
    p.use_graph(True); t.append(time.perf_counter())
    p.set_x(X); t.append(time.perf_counter())
    p.covariance(); p.sync(); t.append(time.perf_counter())
    p.set_w(None); p.sync(); t.append(time.perf_counter())
    p.iterate(20); t.append(time.perf_counter())
    p.sync(); t.append(time.perf_counter())
    out = np.empty((T, F, K), np.complex64); p.demix(True, out=out); t.append(time.perf_counter())
    W = p.get_w(np.complex128); t.append(time.perf_counter())
    p.close(); t.append(time.perf_counter())
    print("complex64 precise: " + " | ".join(f"{n} {ms(x)}" for n, x in zip(("create", "set_precision", "use_graph", "upload", "covariance", "set_w", "queue 20 its", "wait", "demix", "get_w", "close"), np.diff(t)))
          + f" | total {ms(t[-1] - t[0])} ms", flush=True)
    del out
oa.set_precision("precise")
for rep in range(5):
    t0 = time.perf_counter(); Y = oa.overiva(X, n_src=K, n_iter=20); t1 = time.perf_counter()
    print(f"overiva(complex64, precise) 20 its end to end: {ms(t1 - t0)} ms")
    del Y
oa.set_precision("auto")
