"""OGIVE on the GPU (overiva_amd/ive.py -> oiva_plan_ogive_*) against the real reference's recorded outputs
(tests/golden/ogive.npz, from ive.py::ogive) and the oracle.  Entries where the reference itself is chaotic
(amp > 1e3) are not compared.  Needs an MI355X: run with ``-m gpu``."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import ogive_oracle as og
from oracle.overiva_oracle import rel_err, synth_mixture

pytestmark = pytest.mark.gpu
CASES = ("a", "b", "c", "d")
TOL = 1e-5


@pytest.fixture(scope="module")
def gold():
    with np.load(os.path.join(GOLDEN_DIR, "ogive.npz")) as d:
        return {k: d[k] for k in d.files}


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("update", og.UPDATES)
@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_matches_the_reference(gold, case, update, model):
    import overiva_amd as oa

    X = gold[f"{case}_X"].astype(np.complex128)
    checked = 0
    for n_iter in (1, 5, 20, 200):
        key = f"{case}_{update}_{model}_{n_iter}"
        if f"W_{key}" not in gold or float(gold[f"amp_{key}"]) > 1e3:
            continue
        amp = max(1.0, float(gold[f"amp_{key}"]))
        # floor: distance of the reference's OWN complex64 run from its complex128 run (noise of float32 arithmetic
        # inside the loop, which the one-off input perturbation behind amp does not see); the device keeps X, the
        # power pass and r in float32 in both modes, so where that floor is large the bound is a tenth of it
        floor = float(gold.get(f"floor_{key}", 0.0))
        Y, w = oa.ogive(X, n_iter=n_iter, tol=0.0, update=update, proj_back=False, model=model, return_filters=True)
        assert Y.shape == (X.shape[0], X.shape[1], 1) and w.shape == (X.shape[1], X.shape[2], 1) and Y.dtype == np.complex128
        e = rel_err(w, gold[f"W_{key}"])
        print(f"\n[ogive] {key}: w err {e:.2e} (amp {amp:.1f}, reference c64 floor {floor:.1e})")
        assert e < max(TOL * max(1.0, amp / 10.0), 0.1 * floor)
        checked += 1
    assert checked >= 1


@pytest.mark.parametrize("case", CASES)
def test_early_stop_callback_init(gold, case):
    import overiva_amd as oa

    X = gold[f"{case}_X"].astype(np.complex128)
    got = []
    Y = oa.ogive(X, n_iter=400, tol=2e-2, proj_back=True, callback=lambda y: got.append(np.array(y)))
    assert len(got) == int(gold[f"{case}_ncb"])                           # the loop left at the same epoch
    assert rel_err(got[0], gold[f"{case}_cb0"]) < TOL
    if float(gold[f"amp_{case}_demix_laplace_200"]) < 1e3:               # (case b: the reference itself is chaotic by then)
        assert rel_err(Y, gold[f"{case}_Ytol"]) < 10 * TOL
    _, w = oa.ogive(X, n_iter=30, proj_back=False, init_eig=True, return_filters=True)
    assert rel_err(np.abs(w), np.abs(gold[f"{case}_Weig"])) < TOL        # eigenvector phase is LAPACK's choice
    _, w = oa.ogive(X, n_iter=30, proj_back=False, W0=gold[f"{case}_W0"], return_filters=True)
    assert rel_err(w, gold[f"{case}_Ww0"]) < TOL


@pytest.mark.parametrize("M,n_iter", [(5, 40), (16, 20)])
def test_reference_sized_problem(M, n_iter):
    """the shape of the reference's sweep (2049 bins x ~160 frames, overiva_sim.py:313-315) with 5 and 16 channels
    (the VALU and the matrix-core covariance kernels), default update, both arithmetic modes, against the oracle.
    The epoch counts stay where the reference itself is well conditioned in float32 (floor = distance of its own
    complex64 run from its complex128 run: 2e-6 after 50 epochs at 5 channels but 7e-3 after 150; 7e-6 after 20
    epochs at 16 channels but 5e-4 after 40)."""
    import overiva_amd as oa

    X = synth_mixture(160, 2049, M, 2, seed=40 + M)
    (Yr, wr) = og.ogive_staged(X, n_iter=n_iter, tol=0.0, proj_back=True, return_filters=True)
    with np.errstate(all="ignore"):
        (_, w64) = og.ogive_faithful(X, n_iter=n_iter, tol=0.0, proj_back=True, return_filters=True)
    floor = rel_err(w64, wr)
    for mode, floors in (("precise", 0.5), ("fast", 6.0)):
        oa.set_precision(mode)
        try:
            Y, w = oa.ogive(X, n_iter=n_iter, tol=0.0, proj_back=True, return_filters=True)
        finally:
            oa.set_precision("auto")
        e_w, e_y = rel_err(w, wr), rel_err(Y, Yr)
        print(f"\n[ogive] 2049 x 160 x {M}, {n_iter} epochs, {mode}: w err {e_w:.2e}, Y err {e_y:.2e} (reference c64 floor {floor:.1e})")
        assert e_w < max(TOL, floors * floor) and e_y < max(TOL, 3 * floors * floor)


def test_errors():
    import overiva_amd as oa

    X = synth_mixture(32, 4, 3, 2, seed=1)
    with pytest.raises(ValueError):
        oa.ogive(X, update="both")
    with pytest.raises(ValueError):
        oa.ogive(X, model="cauchy")
    with pytest.raises(np.linalg.LinAlgError):
        oa.ogive(np.zeros((32, 4, 3), np.complex64), n_iter=2)
