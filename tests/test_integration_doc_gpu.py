"""INTEGRATION.md section 2 -- the ctypes binding a maintainer of the reference would paste into overiva.py -- is executed as
written (only the library path is made absolute) and its overiva() is compared with the oracle: the documented binding is
code that runs, not prose."""
import os
import re

import numpy as np
import pytest

from oracle import overiva_oracle as orc

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5


@pytest.fixture(scope="module")
def binding():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Binding the C ABI"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    lib = os.path.join(REPO, "overiva_amd", "liboveriva_hip.so")
    assert os.path.exists(lib), "liboveriva_hip.so is not built (python -m overiva_amd.build)"
    assert 'C.CDLL("liboveriva_hip.so")' in code
    ns = {}
    exec(compile(code.replace('C.CDLL("liboveriva_hip.so")', f"C.CDLL({lib!r})"), "INTEGRATION.md#2", "exec"), ns)
    return ns["overiva"]


@pytest.mark.parametrize("case", [(120, 33, 4, 2, np.complex64, "laplace"), (96, 20, 4, 2, np.complex128, "laplace"),
                                  (100, 17, 8, 3, np.complex64, "gauss"), (90, 9, 9, 2, np.complex64, "laplace"),
                                  (64, 5, 6, 6, np.complex128, "laplace"), (80, 7, 16, 5, np.complex64, "laplace")],
                         ids=lambda c: "x".join(str(v) for v in c[:4]) + "-" + np.dtype(c[4]).name + "-" + c[5])
def test_documented_binding_matches_the_oracle(binding, case):
    T, F, M, K, dt, model = case
    X = orc.synth_iid(T, F, M, seed=11).astype(dt)
    Y, W = binding(X, n_src=K, n_iter=7, proj_back=True, model=model, return_filters=True)
    Yr, Wr = orc.overiva_staged(X.astype(np.complex64), n_src=K, n_iter=7, proj_back=True, model=model, return_filters=True)
    assert Y.dtype == dt and W.dtype == dt and Y.shape == (T, F, K) and W.shape == (F, M, K)
    eY, eW = orc.rel_err(Y, Yr), orc.rel_err(W, Wr)
    print(f"\n[INTEGRATION.md binding] {case[:4]} {np.dtype(dt).name} {model}: Y {eY:.1e} W {eW:.1e}")
    assert eY < TOL and eW < TOL


def test_documented_binding_callback_w0_and_eig(binding):
    """overiva.py:142-148 (callback at epochs 0, 10, ... with the projected-back signal), :117 (W0), :106-109 (init_eig)"""
    T, F, M, K = 100, 12, 4, 2
    X = orc.synth_iid(T, F, M, seed=5)
    got, want = [], []
    Y = binding(X, n_src=K, n_iter=12, proj_back=True, callback=lambda y: got.append(np.array(y)))
    Yr = orc.overiva_staged(X, n_src=K, n_iter=12, proj_back=True, callback=lambda y: want.append(np.array(y)))
    assert len(got) == len(want) == 2
    assert orc.rel_err(Y, Yr) < TOL and all(orc.rel_err(a, b) < TOL for a, b in zip(got, want))
    rng = np.random.default_rng(3)
    W0 = (np.eye(M, K)[None] + 0.1 * (rng.standard_normal((F, M, K)) + 1j * rng.standard_normal((F, M, K)))).astype(np.complex64)
    _, W = binding(X, n_src=K, n_iter=3, proj_back=False, W0=W0, return_filters=True)
    _, Wr = orc.overiva_staged(X, n_src=K, n_iter=3, proj_back=False, W0=W0, return_filters=True)
    assert orc.rel_err(W, Wr) < TOL
    Ye = binding(X, n_src=K, n_iter=3, proj_back=True, init_eig=True)
    Yer = orc.overiva_staged(X, n_src=K, n_iter=3, proj_back=True, init_eig=True)
    assert orc.rel_err(Ye, Yer) < TOL


def test_documented_binding_raises_like_the_reference(binding):
    with pytest.raises(KeyError):
        binding(orc.synth_iid(32, 4, 3, seed=1), n_src=2, n_iter=1, model="cauchy")
    with pytest.raises(ValueError):
        binding(orc.synth_iid(32, 4, 3, seed=1), n_src=5, n_iter=1)
