#!/usr/bin/env python3
"""GPU box: where the workgroups of the X-resident kernel are in time -- every workgroup's phase timestamps of one launch:
T F M K [n_iter [mode]].  Prints, per phase boundary, the spread over the workgroups (relative to the earliest) and the
per-phase durations (min / median / max over workgroups), averaged over the iterations after the first two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import overiva_amd as oa
T, F, M, K = [int(a) for a in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 20
mode = sys.argv[6] if len(sys.argv) > 6 else "mixed"
loopback = int(sys.argv[7]) if len(sys.argv) > 7 else 0
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.view_as_complex(torch.randn((T, F, M, 2), generator=g, device="cuda"))
p = oa.Plan(T, F, M, K, "laplace"); p.set_precision(mode); p.set_x_device(X.data_ptr(), X); p.covariance(); p.set_w(None)
if loopback:
    p.resident_loopback(loopback)
p.set_resident(True)
p.iterate(n); p.sync()
p.resident_trace(True)
p.iterate(n); p.sync()
st = p.resident_trace(True, fetch=True).astype(np.int64)       # (wg, iter, 16)
sub = st[:, 2:, :] * 0.01

info = p.resident_info()
NS = info["frame_splits"]
names = ["start", "power", "parts", "activation", "cov_acc", "cov_reduce", "partials", "update", "wait_w"]
st = st[:, 2:, :9] * 0.01                                       # us
t0 = st[:, :, 0].min(axis=0, keepdims=True)                     # earliest start of the iteration over workgroups
print(f"{T}x{F}x{M}/{K} {mode}: grid {info['bin_groups']}x{NS}, iteration = {np.mean(st[:, 1:, 0].min(axis=0) - st[:, :-1, 0].min(axis=0)):.1f} us")
print("boundary      earliest   median   latest  (us after the earliest workgroup started the iteration, mean over iterations)")
for i, nm in enumerate(names):
    rel = st[:, :, i] - t0
    print(f"  {nm:11s} {rel.min(axis=0).mean():8.1f} {np.median(rel, axis=0).mean():8.1f} {rel.max(axis=0).mean():8.1f}")
print("phase         min   median   max  (us, over workgroups, mean over iterations)")
for i in range(1, 9):
    d = st[:, :, i] - st[:, :, i - 1]
    print(f"  {names[i]:11s} {d.min(axis=0).mean():6.1f} {np.median(d, axis=0).mean():6.1f} {d.max(axis=0).mean():6.1f}")
# which workgroups are late at the end of the power phase?
late = (st[:, :, 1] - t0).mean(axis=1)
order = np.argsort(-late)[:8]
print("latest at the end of the power phase: " + ", ".join(f"wg {w} (g {w // NS}, c {w % NS}) +{late[w]:.1f}" for w in order))

# sub-steps of the update (stamps 10..13, workgroups whose wave 0 updates a bin): after the inverses, after the hand-over
# from the helper wave, after source 0, after source 1 -- relative to the end of the wait for the partials (stamp 6)
names2 = ["partials -> inverses", "inverses -> hand-over", "hand-over -> source 0", "source 0 -> source 1", "source 1 -> end of update"]
pts = [6, 10, 11, 12, 13, 7]
ok = (sub[:, :, 10] > 0).all(axis=1)
if ok.any():
    print("update sub-steps (us, median over the updating workgroups, mean over iterations):")
    for i, nm in enumerate(names2):
        d = sub[ok][:, :, pts[i + 1]] - sub[ok][:, :, pts[i]]
        print(f"  {nm:26s} {np.median(d, axis=0).mean():6.2f}")
