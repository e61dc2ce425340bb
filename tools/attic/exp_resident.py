#!/usr/bin/env python3
"""GPU box: the X-resident iteration against the four-launch path: agreement of W, time per iteration, phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import overiva_amd as oa
from overiva_amd import _lib
from oracle import overiva_oracle as orc

CASES = [(300, 40, 8, 2), (200, 64, 4, 2), (160, 24, 8, 1), (120, 33, 4, 1), (1000, 513, 4, 2), (4000, 256, 8, 2), (4000, 250, 8, 2), (3999, 256, 4, 2)]
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (T, F, M, K) in CASES:
    X = orc.synth_mixture(T, F, M, K, seed=5) if T * F < 600000 else orc.synth_iid(T, F, M, seed=5)
    for model in ("laplace", "gauss"):
        for mname, flags in (("fast", _lib.PREC_FAST), ("upd64", _lib.PREC_UPDATE_F64)):
            res = {}
            for resident in (False, True):
                p = oa.Plan(T, F, M, K, model)
                p.set_precision(flags)
                p.set_x(X); p.covariance(); p.set_w(None)
                if resident:
                    info = p.resident_info()
                    if not info["qualifies"]:
                        print(f"{T}x{F}x{M}/{K}: does not qualify", flush=True); p.close(); break
                    p.set_resident(True)
                n = 10
                p.iterate(n); p.sync()
                W = p.get_w(np.complex128)
                # a second call continues from the state (epochs carry over)
                p.iterate(3); p.sync()
                W2 = p.get_w(np.complex128)
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter(); p.iterate(200); p.sync(); ts.append((time.perf_counter() - t0) / 200)
                dt = min(ts)
                res[resident] = (W, W2, dt)
                if resident:
                    ph, nit = p.resident_phases()
                    info = p.resident_info()
                p.close()
            if True in res:
                e1 = orc.rel_err(res[True][0], res[False][0]); e2 = orc.rel_err(res[True][1], res[False][1])
                print(f"{T}x{F}x{M}/{K} {model:7s} {mname:5s}: W rel diff after 10 its {e1:.1e}, after 13 {e2:.1e} | four-launch {res[False][2]*1e6:6.1f} us/it, "
                      f"resident {res[True][2]*1e6:6.1f} us/it | grid {info['bin_groups']}x{info['frame_splits']} TW {info['frames_per_split']} J {info['frames_per_lane']} "
                      f"JR {info['frames_in_registers']} lds {info['lds_bytes']} fallbacks {info['fallbacks']} code {info['last_give_up_code']:#x} | "
                      + " ".join(f"{k} {v:.1f}" for k, v in ph.items()), flush=True)
