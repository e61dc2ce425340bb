#!/usr/bin/env python3
"""Build container only (needs /root/reference): the REAL reference overiva() timed next to the oracle's
reference-faithful restatement (the CPU baseline bench.py reports on the GPU box, where the reference cannot travel), to
show that they are the same speed.  Per-iteration time = time between the reference's own callbacks of epochs 0 and 10, / 10 (prologue
excluded); complex64 i.i.d. input, proj_back=False, laplace, default BLAS threading.

    python tools/cpu_side_by_side.py            -> profiles/r03_cpu_side_by_side.json
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))

import numpy as np  # noqa: E402

from make_golden import import_reference  # noqa: E402
from oracle import overiva_oracle as orc  # noqa: E402


def per_iteration(fn, X, K):
    """seconds per iteration between the callbacks of epochs 0 and 10 (overiva.py:142-148 calls back every 10 epochs;
    with proj_back=False the payload is a view, so the callback costs nothing): the prologue, whose (T,F,M,M) temporary
    makes the reference's set-up time vary by seconds, stays outside"""
    stamps = []
    fn(X, n_src=K, n_iter=11, proj_back=False, model="laplace", callback=lambda Y: stamps.append(time.perf_counter()))
    assert len(stamps) == 2
    return (stamps[1] - stamps[0]) / 10.0


def main():
    ref, _ = import_reference()
    try:
        from threadpoolctl import threadpool_info

        pools = [{k: i.get(k) for k in ("internal_api", "num_threads", "version")} for i in threadpool_info()]
    except Exception:
        pools = None
    out = {"where": "build container (no GPU)", "host_cpus": os.cpu_count(), "blas": pools, "numpy": np.__version__,
           "protocol": "time between the callbacks of epochs 0 and 10, / 10; complex64 iid input, proj_back=False, laplace, default BLAS threading", "cases": []}
    for (T, F, M, K) in ((1000, 513, 4, 2), (1000, 2048, 8, 2)):
        X = orc.synth_iid(T, F, M, seed=0)
        r = per_iteration(ref.overiva, X, K)
        o = per_iteration(orc.overiva_faithful, X, K)
        Wr = ref.overiva(X, n_src=K, n_iter=2, proj_back=False, model="laplace", return_filters=True)[1]
        Wo = orc.overiva_faithful(X, n_src=K, n_iter=2, proj_back=False, model="laplace", return_filters=True)[1]
        out["cases"].append({"shape": {"frames": T, "bins": F, "mics": M, "sources": K},
                             "reference_s_per_iteration": r, "oracle_faithful_s_per_iteration": o, "ratio": o / r,
                             "W_rel_diff_after_2_iterations": float(orc.rel_err(Wo, Wr))})
        print(out["cases"][-1], flush=True)
    path = os.path.join(REPO, "profiles", "r03_cpu_side_by_side.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
