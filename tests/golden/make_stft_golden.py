#!/usr/bin/env python3
"""Golden vectors for the STFT component: inputs and the oracle's outputs (oracle/stft_oracle.py), frozen.

The functions being replaced are third-party (pyroomacoustics 0.1.23, absent from the reference tree and this
image), so there is no reference implementation to record from: PARITY UNPINNED.  The vectors freeze the oracle,
which tests/test_stft_oracle.py pins independently against scipy.signal.stft and closed forms.

Usage:  python tests/golden/make_stft_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import stft_oracle as so  # noqa: E402

CASES = [("a", 64, 32, 3, 64 * 9 + 5), ("b", 256, 64, 2, 256 * 5 + 100), ("c", 128, 128, 1, 128 * 7)]

out = {}
for name, L, hop, C, n in CASES:
    rng = np.random.default_rng(ord(name))
    x = rng.standard_normal((n, C)).astype(np.float32)
    wa = so.hann(L) if hop < L else None
    ws = so.compute_synthesis_window(wa, hop) if hop < L else None
    X = so.analysis(x, L, hop, wa)
    y = so.synthesis(X, L, hop, ws)
    out.update({f"{name}_x": x, f"{name}_L": L, f"{name}_hop": hop, f"{name}_X": X.astype(np.complex64),
                f"{name}_y": y.astype(np.float32)})
np.savez_compressed(os.path.join(HERE, "stft_small.npz"), **out)
print("wrote", os.path.join(HERE, "stft_small.npz"), os.path.getsize(os.path.join(HERE, "stft_small.npz")), "bytes")
