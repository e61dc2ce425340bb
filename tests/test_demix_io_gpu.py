"""The hand-over of Y to the caller's host array (reference overiva.py:192-204), csrc/plan.hip::demix_to_host + csrc/host_io.hip:
slabs of frames through the pinned ring and the copy threads, the page-locked destination, the one-copy form -- same bits in
every form, for complex64 and complex128 output, dense and pitched destinations, with and without projection back."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import overiva_oracle as orc

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def oa():
    import overiva_amd
    from overiva_amd import _lib

    _lib.load()
    return overiva_amd


@pytest.mark.parametrize("shape", [(130, 21, 4, 2), (257, 40, 8, 3), (64, 7, 16, 16), (1000, 64, 2, 1)])
def test_slabs_give_the_bits_of_one_copy(oa, shape):
    T, F, M, K = shape
    X = orc.synth_mixture(T, F, M, K, seed=4)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision("mixed")
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.iterate(2)
        ref = {(pb, dt): p.demix(pb, dtype=dt) for pb in (False, True) for dt in (np.complex64, np.complex128)}      # small: one copy
        for slab in (1, 5000, 64 << 10):            # one frame per slab ... a few slabs
            p.set_io_slab(slab)
            for (pb, dt), want in ref.items():
                got = p.demix(pb, dtype=dt)
                assert got.dtype == dt and np.array_equal(got, want), (slab, pb, dt)
            # a pitched destination: bins 3 .. 3 + F of a wider array, the rest untouched
            out = np.full((T, F + 9, K), 7 + 7j, np.complex128)
            p.demix(True, out=out, f0=3)
            assert np.array_equal(out[:, 3:3 + F], ref[(True, np.complex128)])
            assert np.all(out[:, :3] == 7 + 7j) and np.all(out[:, 3 + F:] == 7 + 7j)
        p.set_io_slab(0)
        assert np.array_equal(p.demix(True), ref[(True, np.complex64)])


def test_prefault_keeps_the_contents(oa):
    import ctypes as C

    from overiva_amd import _lib

    a = np.arange(3_000_001, dtype=np.float64)[1:]         # not page aligned
    keep = a.copy()
    _lib.check(_lib.load().oiva_host_prefault(C.c_void_p(a.ctypes.data), a.nbytes))
    assert np.array_equal(a, keep)


@pytest.mark.parametrize("mode", ["legacy", "ring", "register"])
def test_every_form_of_the_hand_over_in_the_drop_in_call(mode, tmp_path):
    """overiva() end to end in a process of its own with $OIVA_DEMIX_IO set (read once per process): 8 MB of complex64 and
    16 MB of complex128 output, against the legacy form's bits"""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import overiva_amd as oa\n"
        "from oracle import overiva_oracle as orc\n"
        "X = orc.synth_mixture(1000, 513, 4, 2, seed=1)\n"
        "Y64, W = oa.overiva(X, n_src=2, n_iter=3, return_filters=True)\n"
        "Y128 = oa.overiva(X.astype(np.complex128), n_src=2, n_iter=3)\n"
        "np.savez(%r, Y64=Y64, Y128=Y128, W=W)\n") % (REPO, str(tmp_path / f"{mode}.npz"))
    outs = {}
    for m in ("legacy", mode):
        code_m = code.replace(f"{mode}.npz", f"{m}.npz")
        r = subprocess.run([sys.executable, "-c", code_m], env=dict(os.environ, OIVA_DEMIX_IO=m), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[m] = np.load(tmp_path / f"{m}.npz")
    for k in ("Y64", "Y128", "W"):
        assert np.array_equal(outs[mode][k], outs["legacy"][k]), k
    assert outs[mode]["Y128"].dtype == np.complex128


def test_a_plan_kept_between_calls_gives_the_results_of_a_fresh_one(oa):
    """overiva() keeps large plans of the four-launch path for the next call of the same problem (overiva.py::_plan_cache) and
    the library pools their big device buffers: a second call on OTHER data, a warm start, a callback run and the other model
    must give the bits of calls on fresh plans, and release_cached_buffers() must hand the memory back"""
    import torch

    ov = sys.modules["overiva_amd.overiva"]          # (the package attribute `overiva` is the function, not the module)

    T, F, M, K = 1100, 1000, 8, 3                    # 8.8 M elements: above the caching threshold; 3 sources: four-launch path
    X1, X2 = orc.synth_mixture(T, F, M, K, seed=1), orc.synth_iid(T, F, M, seed=2)
    rng = np.random.default_rng(0)
    W0 = (np.eye(M, K)[None] + 0.1 * (rng.standard_normal((F, M, K)) + 1j * rng.standard_normal((F, M, K)))).astype(np.complex64)

    def calls():
        seen = []
        out = [oa.overiva(X1, n_src=K, n_iter=4),
               oa.overiva(X2, n_src=K, n_iter=3, proj_back=False, return_filters=True),
               oa.overiva(X1, n_src=K, n_iter=12, W0=W0, callback=lambda y: seen.append(y.copy())),
               oa.overiva(X2, n_src=K, n_iter=2, model="gauss")]
        return out, seen

    oa.release_cached_buffers()
    os.environ["OIVA_PLAN_CACHE"] = "1"
    a, seen_a = calls()
    assert len(ov._plan_cache) >= 1                   # something was kept
    a2, _ = calls()                                   # every call now starts from a kept plan
    free_held = torch.cuda.mem_get_info()[0]
    oa.release_cached_buffers()
    assert not ov._plan_cache and torch.cuda.mem_get_info()[0] > free_held + (64 << 20)
    keep, ov._PLAN_CACHE_MAX = ov._PLAN_CACHE_MAX, 0  # fresh plans
    try:
        b, seen_b = calls()
    finally:
        ov._PLAN_CACHE_MAX = keep
        oa.release_cached_buffers()
    flat = lambda r: [x for y in r for x in (y if isinstance(y, tuple) else (y,))]
    for x, y, z in zip(flat(a), flat(a2), flat(b)):
        assert np.array_equal(x, z) and np.array_equal(y, z)
    assert len(seen_a) == len(seen_b) == 2 and all(np.array_equal(p, q) for p, q in zip(seen_a, seen_b))


def test_graph_cache_eviction_pool_trim_and_replanning_keep_the_results(oa):
    """one plan driven through more graph lengths than its cache holds (6), a trim of the buffer pool in the middle, a new X on
    the same plan and a precision switch: every state it passes through equals that of a plan driven eagerly"""
    T, F, M, K = 300, 200, 4, 2
    X1, X2 = orc.synth_iid(T, F, M, seed=5), orc.synth_mixture(T, F, M, K, seed=6)
    chunks = [1, 2, 3, 5, 7, 11, 13, 2, 40, 1]          # 9 distinct lengths (40 = 32 + 8)

    def run(graph):
        out = []
        with oa.Plan(T, F, M, K, "laplace") as p:
            p.set_precision("mixed")
            p.use_graph(graph)
            for X in (X1, X2):
                p.set_x(X)
                p.covariance()
                p.set_w(None)
                for i, n in enumerate(chunks):
                    p.iterate(n)
                    if i == 4:
                        oa.release_cached_buffers()
                out.append(p.get_w(np.complex128))
                out.append(p.demix(True))
            p.set_precision("fast")
            p.iterate(3)
            out.append(p.get_w())
        return out

    for a, b in zip(run(False), run(True)):
        assert np.array_equal(a, b)


def test_a_kept_plan_returns_the_iterated_w_in_fast_determined(oa):
    """with `fast` (float32 per-bin algebra: the complex128 copy of W_hat is not maintained) and n_src == n_chan
    (no J stage that would clear the flag), the second call of a shape replays the CACHED graph -- nothing is captured, so the
    host flag "the complex128 copy is current" stayed true from the upload of W0 and get_w returned W0 = identity.  Two calls on
    different data, each against a fresh plan."""
    ov = sys.modules["overiva_amd.overiva"]
    T, F, M = 1100, 1000, 8                            # 8.8 M elements: graphs and the plan cache apply
    X1, X2 = orc.synth_iid(T, F, M, seed=11), orc.synth_mixture(T, F, M, M, seed=12)
    oa.release_cached_buffers()
    oa.set_precision("fast")
    try:
        kept = [oa.overiva(X, n_iter=3, proj_back=False, return_filters=True) for X in (X1, X2, X1)]
        assert len(ov._plan_cache) >= 1
        oa.release_cached_buffers()
        keep, ov._PLAN_CACHE_MAX = ov._PLAN_CACHE_MAX, 0
        try:
            fresh = [oa.overiva(X, n_iter=3, proj_back=False, return_filters=True) for X in (X1, X2, X1)]
        finally:
            ov._PLAN_CACHE_MAX = keep
    finally:
        oa.set_precision("auto")
        oa.release_cached_buffers()
    eye = np.broadcast_to(np.eye(M, dtype=np.complex64), (F, M, M))
    for (Yk, Wk), (Yf, Wf) in zip(kept, fresh):
        assert np.array_equal(Yk, Yf) and np.array_equal(Wk, Wf)
        assert not np.allclose(Wk, eye)                # (what the stale flag returned)
