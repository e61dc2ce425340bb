"""The activation floor ``r[r < eps] = eps`` (reference overiva.py:170-173), the reference's one guarded edge case.

Fixtures ``tests/golden/floor_*.npz`` (make_floor_golden.py, outputs of the REAL reference): inputs with frames of exact
zeros -- one stretch longer than a frame split of the X-resident kernel --, frames scaled by 1e-8 / 3e-8 (gauss: floored,
and the floored weight 1e15 still gives them about a tenth of a normal frame's share of V: WHERE the floor sits shows in
W), by 1e-12 and by 1e-20.  CPU: the oracle against them.  GPU: the four-launch path and the X-resident kernel (whose
weights are formed before gamma is known: it corrects the floored frames once gamma has arrived) against them, in the
arithmetic of complex64 and of complex128 input.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import overiva_oracle as orc

FLOOR_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "floor_*.npz")))
FLOOR_IDS = [os.path.basename(p)[len("floor_"):-len(".npz")] for p in FLOOR_FILES]
TOL = 1e-5


@pytest.fixture(params=FLOOR_FILES, ids=FLOOR_IDS)
def floor_case(request):
    with np.load(request.param) as d:
        return {k: d[k] for k in d.files}


def test_fixtures_exist():
    assert len(FLOOR_FILES) >= 3


@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_oracle_reproduces_the_reference_floor(floor_case, model):
    """the restatement against the real reference on inputs where the floor fires in every epoch: complex128 to 1e-9,
    complex64 as close as the two complex64 runs are to the complex128 one"""
    g = floor_case
    X, K = g["X"], int(g["K"])
    assert g[f"floored_{model}"].min() > 0
    for n in (1, 5):
        _, W = orc.overiva_faithful(X.astype(np.complex128), n_src=K, n_iter=n, proj_back=False, model=model, return_filters=True)
        assert orc.rel_err(W, g[f"W_c128_{model}_{n}"]) < 1e-9
        _, W = orc.overiva_faithful(X, n_src=K, n_iter=n, proj_back=False, model=model, return_filters=True)
        floor = orc.rel_err(g[f"W_c64_{model}_{n}"], g[f"W_c128_{model}_{n}"])
        assert orc.rel_err(W, g[f"W_c64_{model}_{n}"]) < max(2e-4, 4 * floor)
        _, W = orc.overiva_staged(X, n_src=K, n_iter=n, proj_back=False, model=model, return_filters=True)
        assert orc.rel_err(W, g[f"W_c128_{model}_{n}"]) < 1e-5


def test_the_floor_matters_in_these_fixtures(floor_case):
    """without the floor (or with it in another place) the gauss result differs by far more than the tolerance the GPU
    paths are held to: the fixtures can tell"""
    g = floor_case
    X, K = g["X"].astype(np.complex128), int(g["K"])
    keep = orc.EPS_R
    try:
        orc.EPS_R = 1e-20
        _, W = orc.overiva_faithful(X, n_src=K, n_iter=1, proj_back=False, model="gauss", return_filters=True)
    finally:
        orc.EPS_R = keep
    assert orc.rel_err(W, g["W_c128_gauss_1"]) > 1e-3


# ---- GPU -------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oa():
    import overiva_amd
    from overiva_amd import _lib

    _lib.load()
    return overiva_amd


def _run(oa, X, K, model, mode, n_iter, resident):
    T, F, M = X.shape
    with oa.Plan(T, F, M, K, model) as p:
        p.set_precision(mode)
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        if resident:
            p.set_resident(True)
        p.iterate(n_iter)
        W = p.get_w(np.complex128)
        info = p.resident_info()
    return W, info


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["four_launch", "resident"])
@pytest.mark.parametrize("mode", ["fast", "mixed", "precise"])
@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_floor_on_the_gpu(oa, floor_case, model, mode, path):
    """both paths, every arithmetic mode, against the REAL reference's W on inputs with silent and nearly silent frames"""
    g = floor_case
    X, K = g["X"], int(g["K"])
    resident = path == "resident"
    if resident and mode == "precise" and X.shape[2] != 4:
        pytest.skip("the float64 covariance exists in the X-resident kernel for 4 channels only")
    for n in (1, 5):
        W, info = _run(oa, X, K, model, mode, n, resident)
        if resident:
            assert info["enabled"] == 1 and info["fallbacks"] == 0 and info["launches"] == 1
        ref128, ref64 = g[f"W_c128_{model}_{n}"], g[f"W_c64_{model}_{n}"]
        floor = orc.rel_err(ref64, ref128)
        e128, e64 = orc.rel_err(W, ref128), orc.rel_err(W, ref64)
        print(f"\n[floor] {path} {mode} {model} n={n}: vs reference c128 {e128:.1e}, vs reference c64 {e64:.1e} (its own floor {floor:.1e})")
        assert np.all(np.isfinite(W))
        assert e128 < TOL and e64 < max(TOL, 1.5 * floor)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_floor_resident_equals_four_launch(oa, floor_case, model):
    """the two paths place gamma differently (the X-resident kernel forms its weights before gamma is known and corrects the
    floored frames afterwards): same W to 1e-6 on inputs where the floor fires in every iteration, 12 iterations"""
    g = floor_case
    X, K = g["X"], int(g["K"])
    Wr, info = _run(oa, X, K, model, "mixed", 12, True)
    W4, _ = _run(oa, X, K, model, "mixed", 12, False)
    assert info["fallbacks"] == 0
    assert orc.rel_err(Wr, W4) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_floor_through_overiva(oa, floor_case, model):
    """the drop-in call (it picks the X-resident kernel where the shape qualifies), complex64 and complex128 input"""
    g = floor_case
    K = int(g["K"])
    for dt, key in ((np.complex64, "c64"), (np.complex128, "c128")):
        Y, W = oa.overiva(g["X"].astype(dt), n_src=K, n_iter=5, proj_back=False, model=model, return_filters=True)
        assert W.dtype == dt and np.all(np.isfinite(Y))
        floor = orc.rel_err(g[f"W_c64_{model}_5"], g[f"W_c128_{model}_5"])
        assert orc.rel_err(W, g[f"W_{key}_{model}_5"]) < max(TOL, 1.5 * floor)
