#!/usr/bin/env python3
"""
Golden vectors for the activation floor ``r[r < eps] = eps`` (reference overiva.py:170-171).  Build container only
(imports the real /root/reference/overiva.py through the shims of make_golden.py; the reference is not modified).

None of the other fixtures has a frame whose normalised activation is below 1e-15, so the one edge case the reference
guards against never fired in a test.  These inputs do: frames of exact zeros (digital silence; one stretch is longer than
a frame split of the X-resident kernel), frames scaled by 1e-8 / 3e-8 (gauss: r / gamma ~ 1e-16 -- floored, and the
floored weight 1e15 still gives them a tenth of a normal frame's share of V, so WHERE the floor sits is visible in W),
by 1e-12 (gauss: floored, negligible share; laplace: not floored) and by 1e-20 (floored in both models; |y|^2 is a
float32 denormal).  Stored: X, the frame lists, and the reference's W for {complex64, complex128} x {laplace, gauss} x
n_iter in {1, 5}, plus how many (frame, source) pairs the floor changed in the complex128 run (traced at overiva.py:173).

Usage:  python tests/golden/make_floor_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from make_golden import import_reference  # noqa: E402
from oracle.overiva_oracle import synth_iid  # noqa: E402

OUT = os.path.join(HERE, "floor_{name}.npz")
# (name, T, F, M, K): both shapes qualify for the X-resident kernel (4 / 8 channels, 1-2 sources + background)
CASES = [("m4", 160, 32, 4, 2), ("m8", 176, 32, 8, 2), ("m8k1", 96, 24, 8, 1)]
N_ITERS = (1, 5)


def silence(X):
    """scale frames of X in place; returns {scale: frames}"""
    T = X.shape[0]
    plan = {0.0: list(range(3, 6)) + list(range(64, 64 + 20)), 1e-8: [40, 41], 3e-8: [60], 1e-12: [20, T - 1], 1e-20: [21, 22]}
    for s, frames in plan.items():
        X[frames] *= np.float32(s)
    return plan


def count_floored(ref_overiva, X, K, n_iter, model):
    """(frame, source) pairs with r_inv == 1 / eps when overiva.py:176 is reached, per epoch"""
    hits = []
    target = ref_overiva.overiva.__code__

    def tracer(frame, event, arg):
        if frame.f_code is not target:
            return None

        def local(frame, event, arg):
            if event == "line" and frame.f_lineno == 176:
                hits.append(int(np.sum(frame.f_locals["r_inv"] == 1.0 / 1e-15)))
            return local

        return local

    sys.settrace(tracer)
    try:
        ref_overiva.overiva(X, n_src=K, n_iter=n_iter, proj_back=False, model=model)
    finally:
        sys.settrace(None)
    return hits


def main():
    ref_overiva, _ = import_reference()
    for name, T, F, M, K in CASES:
        X = synth_iid(T, F, M, seed=4242 + M + K)
        plan = silence(X)
        out = {"X": X, "K": K}
        for s, frames in plan.items():
            out[f"frames_{s:g}"] = np.array(frames)
        for model in ("laplace", "gauss"):
            out[f"floored_{model}"] = np.array(count_floored(ref_overiva, X.astype(np.complex128), K, max(N_ITERS), model))
            assert out[f"floored_{model}"].min() > 0
            for dt in ("c64", "c128"):
                Xd = X if dt == "c64" else X.astype(np.complex128)
                for n in N_ITERS:
                    Y, W = ref_overiva.overiva(Xd.copy(), n_src=K, n_iter=n, proj_back=False, model=model, return_filters=True)
                    assert np.all(np.isfinite(W)) and W.dtype == Xd.dtype
                    out[f"W_{dt}_{model}_{n}"] = np.ascontiguousarray(W)
            print(name, model, "floored (frame, source) pairs per epoch:", out[f"floored_{model}"].tolist(),
                  "c64 vs c128 W after 5:", float(np.linalg.norm(out[f"W_c64_{model}_5"] - out[f"W_c128_{model}_5"]) /
                                                  np.linalg.norm(out[f"W_c128_{model}_5"])))
        np.savez_compressed(OUT.format(name=name), **out)
        print("wrote", OUT.format(name=name), os.path.getsize(OUT.format(name=name)), "bytes")


if __name__ == "__main__":
    main()
