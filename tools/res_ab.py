#!/usr/bin/env python3
"""GPU box: the X-resident kernel of several builds of the library side by side (tools/build_variant.py):
res_ab.py [variant ...] -- for BASELINE configs[0] / configs[1] and the 8-GPU shard: microseconds per iteration at 20 and
100 iterations per launch and workgroup 0's phases."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("cfg0", 160, 2049, 4, 2, "precise"), ("cfg2", 1000, 513, 4, 2, "mixed"), ("shard8", 4000, 256, 8, 2, "mixed")]
WORKER = r'''
import sys, time
sys.path.insert(0, %r)
import torch, overiva_amd as oa
name, T, F, M, K, mode = sys.argv[1], *[int(a) for a in sys.argv[2:6]], sys.argv[6]
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
out = []
for n in (20, 100):
    p = oa.Plan(T, F, M, K, "laplace"); p.set_precision(mode); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
    p.set_resident(True); p.iterate(10); p.sync()
    ts = []
    for r in range(9):
        t0 = time.perf_counter(); p.iterate(n); p.sync(); ts.append(time.perf_counter() - t0)
    ph, _ = p.resident_phases()
    out.append("%%d its: %%6.2f us" %% (n, sorted(ts)[len(ts) // 2] / n * 1e6))
    fb = p.resident_info()["fallbacks"]
    p.close()
print("%%-7s" %% name, " | ".join(out), "|", " ".join("%%s %%.1f" %% (k[:9], v) for k, v in ph.items()), "| fallbacks", fb)
''' % REPO
for variant in (sys.argv[1:] or [""]):
    lib = os.path.join(REPO, "overiva_amd", f"liboveriva_hip_{variant}.so" if variant else "liboveriva_hip.so")
    print("==", variant or "default", flush=True)
    for c in CASES:
        env = dict(os.environ, OIVA_LIB=lib)
        r = subprocess.run([sys.executable, "-c", WORKER, *[str(x) for x in c]], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-800:], flush=True)
