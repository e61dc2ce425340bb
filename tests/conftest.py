import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs(*keys): entries the golden fixture of this parametrisation must hold; `{model}`-style "
                                       "fields are filled from the test's other parameters.  Parametrisations without them are "
                                       "DESELECTED at collection (the F >= 64 fixtures store W only, for n_iter 1, 5, 20)")


_KEYS = {}


def _fixture_keys(path):
    if path not in _KEYS:
        with np.load(path) as d:
            _KEYS[path] = set(d.files)
    return _KEYS[path]


def pytest_collection_modifyitems(config, items):
    keep, drop = [], []
    for it in items:
        m = it.get_closest_marker("needs")
        cs = getattr(it, "callspec", None)
        if m is not None and cs is not None and "golden" in cs.params:
            have = _fixture_keys(cs.params["golden"])
            fields = {k: v for k, v in cs.params.items() if k != "golden"}
            if any(k.format(**fields) not in have for k in m.args):
                drop.append(it)
                continue
        keep.append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "overiva_*.npz")))


def golden_ids():
    return [os.path.basename(p)[len("overiva_"):-len(".npz")] for p in golden_files()]


@pytest.fixture(params=golden_files(), ids=golden_ids())
def golden(request):
    """one committed fixture = inputs + outputs of the real reference (tests/golden/make_golden.py)"""
    with np.load(request.param) as d:
        g = {k: d[k] for k in d.files}
    g["_id"] = os.path.basename(request.param)[len("overiva_"):-len(".npz")]
    return g


def need(g, *keys):
    """skip when a fixture does not hold an entry (tests that cannot say so with @pytest.mark.needs)"""
    for k in keys:
        if k not in g:
            pytest.skip(f"fixture {g['_id']} has no {k}")


AMP_LIMIT = 1e3


def chaotic(g, model, n_iter):
    """True when the REFERENCE ITSELF is ill-conditioned for this (fixture, model, n_iter).

    make_golden.py stores ``amp_<model>_<n_iter>`` = relative change of the real reference's
    complex128 result under a 1e-12 relative perturbation of X, divided by 1e-12.  The laplace
    model stays at amp ~ 1..10 everywhere; the gauss model on these small-F fixtures blows up to
    1e7..1e12 between 5 and 20 iterations (an activation r = sum_f |y|^2 / F over a handful of
    bins gets arbitrarily close to 0).  Nothing can be pinned tighter than amp * rounding, so
    such entries are skipped rather than compared at a meaningless tolerance."""
    key = f"amp_{model}_{n_iter}"
    return key in g and float(g[key]) > AMP_LIMIT


JITTER_LIMIT = 1e-3
_JITTER = None


def c64_jitter(g, model, n_iter):
    """largest relative change of the REAL reference's complex64 W on this (fixture, model, n_iter) when every sample of X
    moves by one unit in the last place (tests/golden/make_jitter_golden.py -> c64_jitter.npz), or None"""
    global _JITTER
    if _JITTER is None:
        path = os.path.join(GOLDEN_DIR, "c64_jitter.npz")
        _JITTER = {}
        if os.path.exists(path):
            with np.load(path) as d:
                _JITTER = {k: float(d[k]) for k in d.files}
    return _JITTER.get(f"{g['_id']}_{model}_{n_iter}")


def c64_diverged(g, model, n_iter):
    """True when the reference's OWN complex64 run is not reproducible to 1e-3 under a last-bit change of its input (measured
    on the reference, c64_jitter above): the complex64 counterpart of `chaotic`.  Four non-chaotic rows: e_mix laplace 20
    (jitter 7.1e-3, floor 1.8e-3), l_mix gauss 5 (6.1e-3 / 3.7e-3), w_mix gauss 20 (5.7e-3 / 2.7e-3), v_mix gauss 20 (2.2e-3).  On such a row the
    distance from the reference's complex64 result is held to its jitter instead of its floor."""
    j = c64_jitter(g, model, n_iter)
    return j is not None and j > JITTER_LIMIT


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
