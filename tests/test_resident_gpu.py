"""The X-resident iteration (csrc/resident_kernel.inc): the loop body of reference overiva.py:138-190 as one persistent
launch with X on chip.  Checked against the reference's own outputs (golden fixtures), against the oracle at the sizes
it exists for (BASELINE configs[1], one rank's 256-bin shard of the headline shape), against the four-launch path, and
for what happens when its workgroups cannot all run (it must give up, change nothing and fall back)."""
import numpy as np
import pytest

from conftest import chaotic, golden_files, golden_ids, need  # noqa: F401
from oracle import overiva_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def oa():
    import overiva_amd
    from overiva_amd import _lib

    _lib.load()
    return overiva_amd


def _run(oa, X, K, model, mode, n_iter, resident, W0=None, chunks=None):
    T, F, M = X.shape
    with oa.Plan(T, F, M, K, model) as p:
        p.set_precision(mode)
        p.set_x(X)
        p.covariance()
        p.set_w(W0)
        if resident:
            p.set_resident(True)
        for n in (chunks or [n_iter]):
            p.iterate(n)
        W = p.get_w(np.complex128)
        Y = p.demix(False)
        info = p.resident_info()
    return W, Y, info


def _qualifying():
    out = []
    for path, gid in zip(golden_files(), golden_ids()):
        with np.load(path) as d:
            M, K = d["X"].shape[2], int(d["K"])
        if M in (2, 4, 6, 8) and K in (1, 2) and K < M:
            out.append(pytest.param(path, id=gid))
    return out


@pytest.mark.parametrize("mode", ["fast", "mixed"])
@pytest.mark.parametrize("model", ["laplace", "gauss"])
@pytest.mark.parametrize("path", _qualifying())
def test_resident_matches_reference_fixtures(oa, path, model, mode):
    """final W after 1, 5 and 20 iterations against the REAL reference's outputs: complex128 result within the bound of the
    mode (mixed: the north-star 1e-5 or 1.5 of the reference's own complex64 floors), and never further from it than
    the four-launch path of the same mode by more than a floor"""
    with np.load(path) as d:
        g = {k: d[k] for k in d.files}
    g["_id"] = path
    X, K = g["X"].astype(np.complex64), int(g["K"])
    checked = 0
    for n in (1, 5, 20):
        k128, k64 = f"W_c128_{model}_{n}", f"W_c64_{model}_{n}"
        if k128 not in g or k64 not in g or chaotic(g, model, n):
            continue
        W, _, info = _run(oa, X, K, model, mode, n, True)
        assert info["enabled"] == 1 and info["fallbacks"] == 0 and info["launches"] >= 1
        W4, _, _ = _run(oa, X, K, model, mode, n, False)
        ref, floor = g[k128], orc.rel_err(g[k64], g[k128])
        e, e4 = orc.rel_err(W, ref), orc.rel_err(W4, ref)
        print(f"\n[resident] {path[-10:-4]} {model} {mode} n={n}: resident {e:.1e}, four-launch {e4:.1e}, reference c64 floor {floor:.1e}")
        bound = max(TOL, (1.5 if mode == "mixed" else 6.0) * floor)
        assert e < bound
        assert e < e4 + max(floor, 1e-6)
        checked += 1
    assert checked > 0


@pytest.mark.parametrize("shape", [(1000, 513, 4, 2), (4000, 256, 8, 2), (4000, 250, 8, 1), (3999, 256, 4, 2), (700, 96, 8, 2),
                                   (160, 2049, 4, 2), (235, 2049, 4, 1), (200, 800, 8, 2), (120, 1000, 8, 1),
                                   # (round 4) 6 and 2 channels: 2 channels at the reference's 2049 bins (6 and 8 channels update one bin
                                   # per wave, i.e. need >= 4 frame splits: at most 1024 bins), register frames at 6 channels, ragged
                                   # bin groups
                                   (235, 1024, 6, 2), (160, 1000, 6, 1), (235, 2049, 2, 1), (4000, 200, 6, 2), (500, 70, 2, 1)])
@pytest.mark.parametrize("mode", ["fast", "mixed"])
def test_resident_equals_four_launch_path(oa, shape, mode):
    """BASELINE configs[1], one rank's shard of the headline shape at 8 GPUs (full and ragged), 4 channels with 16 frames
    per lane in LDS, a small-split shape, and shapes with 40 and more bin groups -- the reference's own 2049 bins among them
    -- whose powers are exchanged in two hops: same W and Y as the four-launch path to 1e-6 (the two differ in where gamma is
    applied and in one mantissa bit of the exchanged parts), also when the iterations come in several calls"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=3)
    for model in ("laplace", "gauss"):
        Wr, Yr, info = _run(oa, X, K, model, mode, 12, True, chunks=[5, 1, 6])
        W4, Y4, _ = _run(oa, X, K, model, mode, 12, False)
        assert info["fallbacks"] == 0 and info["launches"] == 3
        eW, eY = orc.rel_err(Wr, W4), orc.rel_err(Yr, Y4)
        print(f"\n[resident] {shape} {model} {mode}: W {eW:.1e} Y {eY:.1e} grid {info['bin_groups']}x{info['frame_splits']} "
              f"frames/lane {info['frames_per_lane']} ({info['frames_in_registers']} in registers)")
        assert eW < 1e-6 and eY < 1e-6


@pytest.mark.parametrize("mode", ["fast", "mixed"])
def test_resident_cfg2_against_oracle(oa, mode):
    """BASELINE configs[1] (513 x 1000 x 4 / 2, laplace), iid and mixture, 10 iterations against the oracle"""
    T, F, M, K = 1000, 513, 4, 2
    for name, X in (("iid", orc.synth_iid(T, F, M, seed=0)), ("mix", orc.synth_mixture(T, F, M, K, seed=2))):
        Yr, Wr = orc.overiva_staged(X.astype(np.complex128), n_src=K, n_iter=10, proj_back=False, return_filters=True)
        W, Y, info = _run(oa, X, K, "laplace", mode, 10, True)
        eW, eY = orc.rel_err(W, Wr), orc.rel_err(Y, Yr)
        print(f"\n[resident] cfg2 {name} {mode}: W {eW:.1e} Y {eY:.1e}")
        assert info["fallbacks"] == 0
        bound = TOL if name == "iid" or mode == "mixed" else 6e-5      # (fast on a mixture: the envelope of that mode)
        assert eW < bound and eY < bound


def test_resident_shard_against_oracle(oa):
    """one rank's shard of the headline shape at 8 GPUs (256 x 4000 x 8 / 2): 3 iterations against the oracle"""
    T, F, M, K = 4000, 256, 8, 2
    X = orc.synth_iid(T, F, M, seed=1)
    Yr, Wr = orc.overiva_staged(X.astype(np.complex128), n_src=K, n_iter=3, proj_back=False, return_filters=True)
    for mode in ("fast", "mixed"):
        W, Y, info = _run(oa, X, K, "laplace", mode, 3, True)
        assert info["fallbacks"] == 0 and info["frames_in_registers"] == 8 and info["bin_groups"] * info["frame_splits"] == 256
        assert orc.rel_err(W, Wr) < TOL and orc.rel_err(Y, Yr) < TOL


@pytest.mark.parametrize("shape", [(1000, 513, 4, 2), (160, 2049, 4, 2), (300, 64, 4, 1), (77, 130, 4, 2)])
def test_resident_precise_4_channels(oa, shape):
    """`precise` (float64 covariance sums of exact products, the arithmetic of complex128 input) inside the kernel at 4
    channels -- the reference's own case (BASELINE configs[0]: 2049 bins x ~160 frames x 4 mics): same W and Y as the
    four-launch `precise` path to 1e-6 (the two differ in where gamma is applied and in one mantissa bit of the exchanged
    float32 powers, like the float32 forms), iterations in several calls; and on a mixture both within 1e-5 of the oracle's
    complex128 result"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=5)
    for model in ("laplace", "gauss"):
        Wr, Yr, info = _run(oa, X, K, model, "precise", 12, True, chunks=[5, 1, 6])
        W4, Y4, _ = _run(oa, X, K, model, "precise", 12, False)
        assert info["fallbacks"] == 0 and info["launches"] == 3
        eW, eY = orc.rel_err(Wr, W4), orc.rel_err(Yr, Y4)
        print(f"\n[resident] precise {shape} {model}: W {eW:.1e} Y {eY:.1e}")
        assert eW < 1e-6 and eY < 1e-6
    Xm = orc.synth_mixture(T, F, M, K, seed=5)
    Yo, Wo = orc.overiva_staged(Xm.astype(np.complex128), n_src=K, n_iter=10, proj_back=False, return_filters=True)
    Wr, Yr, info = _run(oa, Xm, K, "laplace", "precise", 10, True)
    assert info["fallbacks"] == 0 and info["launches"] == 1
    assert orc.rel_err(Wr, Wo) < TOL and orc.rel_err(Yr, Yo) < TOL


def test_precise_stays_on_four_launches_at_8_channels(oa):
    """128 float64 accumulators do not fit next to X: `precise` with 8 channels keeps the four-launch path even when the
    switch is on"""
    T, F, M, K = 300, 64, 8, 2
    X = orc.synth_iid(T, F, M, seed=1)
    _, _, info = _run(oa, X, K, "laplace", "precise", 3, True)
    assert info["launches"] == 0 and info["fallbacks"] == 0


def test_resident_gives_up_and_falls_back(oa):
    """a workgroup that never publishes (test hook) = what a grid that is not resident as a whole looks like: every wait
    runs into its time-out, the launch returns without having written W, the plan reports why and runs the call -- and the
    following ones -- on the four-launch path with the same result as if resident had never been on"""
    T, F, M, K = 300, 64, 4, 2
    X = orc.synth_mixture(T, F, M, K, seed=9)
    W4, Y4, _ = _run(oa, X, K, "laplace", "mixed", 7, False)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision("mixed")
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.set_resident(True)
        p.resident_debug(timeout_ms=20, stall_block=3)
        p.iterate(4)
        info = p.resident_info()
        assert info["fallbacks"] == 1 and info["enabled"] == 0 and info["last_give_up_code"] != 0
        p.resident_debug(0, -1)
        p.iterate(3)
        assert p.resident_info()["launches"] == 1
        W = p.get_w(np.complex128)
        Y = p.demix(False)
        assert np.array_equal(W, W4) and np.array_equal(Y, Y4)
        # turned on again it works (buffers and epochs were reset)
        p.set_resident(True)
        p.iterate(2)
        assert p.resident_info()["fallbacks"] == 1 and p.resident_info()["launches"] == 2
        assert np.all(np.isfinite(p.get_w()))


@pytest.mark.parametrize("shape,mode", [((300, 64, 4, 2), "mixed"), ((700, 96, 8, 2), "fast"), ((120, 700, 4, 2), "mixed")])
def test_give_up_in_the_last_iteration_changes_nothing(oa, shape, mode):
    """a time-out that first strikes in the LAST iteration of a launch: most workgroups have finished all their iterations by
    then and have written their final W_hat -- to the staging copy, which the host moves into place only when nobody gave up.
    The call must leave W_hat as it was and run all its iterations on the four-launch path: same bits as that path alone
    (a write-back from inside the kernel would leave the finished bins with twice the iterations)"""
    T, F, M, K = shape
    X = orc.synth_mixture(T, F, M, K, seed=4)
    W4, Y4, _ = _run(oa, X, K, "laplace", mode, 6, False)
    for stall_from in (5, 3):
        with oa.Plan(T, F, M, K, "laplace") as p:
            p.set_precision(mode)
            p.set_x(X)
            p.covariance()
            p.set_w(None)
            p.set_resident(True)
            p.resident_debug(timeout_ms=20, stall_block=1, from_iteration=stall_from)
            p.iterate(6)
            info = p.resident_info()
            assert info["fallbacks"] == 1 and info["enabled"] == 0 and info["last_give_up_code"] != 0
            assert np.array_equal(p.get_w(np.complex128), W4) and np.array_equal(p.demix(False), Y4)


@pytest.mark.parametrize("shape", [(4000, 256, 8, 2), (1000, 513, 4, 2), (700, 96, 8, 1), (300, 64, 4, 2)])
@pytest.mark.parametrize("world", [8, 2])
def test_loopback_world_runs_the_multi_gpu_exchange_on_one_gpu(oa, shape, world):
    """the multi-GPU code path of the kernel (per frame split a leader gathers the rank's parts and stores the sums of all
    `world` slots into the gather buffer, every workgroup polls its words of all slots and adds them in rank order) with
    the one GPU playing every rank: the phantom ranks' sums are exact zeros, so the result is the single-rank one -- to the
    mantissa bit the epoch tag takes from the stored sums"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=8)
    for model in ("laplace", "gauss"):
        W1, Y1, _ = _run(oa, X, K, model, "mixed", 7, True, chunks=[3, 4])
        with oa.Plan(T, F, M, K, model) as p:
            p.set_precision("mixed")
            p.set_x(X)
            p.covariance()
            p.set_w(None)
            p.resident_loopback(world)
            p.set_resident(True)
            p.iterate(3)
            p.iterate(4)
            info = p.resident_info()
            W, Y = p.get_w(np.complex128), p.demix(False)
            assert info["fallbacks"] == 0 and info["launches"] == 2
            # and off again: the plain single-rank kernel
            p.set_resident(False)
            p.resident_loopback(0)
            p.set_resident(True)
            p.iterate(1)
            assert p.resident_info()["fallbacks"] == 0 and np.all(np.isfinite(p.get_w()))
        eW, eY = orc.rel_err(W, W1), orc.rel_err(Y, Y1)
        print(f"\n[loop-back world {world}] {shape} {model}: W {eW:.1e} Y {eY:.1e} vs the single-rank kernel")
        assert eW < 1e-6 and eY < 1e-6


def test_loopback_gives_up_like_a_missing_rank(oa):
    """a leader that never stores its slots = a rank whose parts do not arrive: the launch gives up and reports it (a plan
    with world > 1 does not fall back silently -- the other ranks' state is unknown)"""
    T, F, M, K = 300, 64, 4, 2
    X = orc.synth_iid(T, F, M, seed=8)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision("mixed")
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.resident_loopback(8)
        p.set_resident(True)
        W0 = p.get_w(np.complex128)
        p.resident_debug(timeout_ms=20, stall_block=0)
        with pytest.raises(RuntimeError):
            p.iterate(3)
        assert np.array_equal(p.get_w(np.complex128), W0)


def test_two_hop_exchange_gives_up_too(oa):
    """the same with 44 bin groups (the reduce-scatter / all-gather exchange of the powers has its own waits): a workgroup
    that never publishes its share of the column's sums stalls every workgroup of the column; the launch gives up, changes
    nothing, and the plan continues on the four-launch path"""
    T, F, M, K = 120, 700, 4, 2
    X = orc.synth_iid(T, F, M, seed=9)
    W4, Y4, _ = _run(oa, X, K, "laplace", "mixed", 5, False)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision("mixed")
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.set_resident(True)
        assert p.resident_info()["bin_groups"] >= 40
        p.resident_debug(timeout_ms=20, stall_block=1)
        p.iterate(5)
        info = p.resident_info()
        assert info["fallbacks"] == 1 and info["enabled"] == 0 and info["last_give_up_code"] != 0
        assert np.array_equal(p.get_w(np.complex128), W4) and np.array_equal(p.demix(False), Y4)


def test_shapes_that_do_not_qualify(oa):
    for shape in ((4000, 2048, 8, 2), (200, 40, 7, 2), (200, 40, 8, 8), (200, 40, 8, 3), (200, 40, 6, 3), (200, 40, 2, 2)):
        T, F, M, K = shape
        with oa.Plan(T, F, M, K, "laplace") as p:
            assert p.resident_info()["qualifies"] == 0
            with pytest.raises(ValueError):
                p.set_resident(True)


def test_overiva_uses_the_resident_kernel_when_it_applies(oa):
    """the drop-in call on a qualifying complex64 input: callback cadence, projection back and the result are those of
    the reference (golden fixture h: 200 x 64 x 4 / 2)"""
    import os

    path = [p for p in golden_files() if p.endswith("h_mix.npz")][0]
    with np.load(path) as d:
        g = {k: d[k] for k in d.files}
    X, K = g["X"].astype(np.complex64), int(g["K"])
    seen = []
    Y, W = oa.overiva(X, n_src=K, n_iter=20, proj_back=False, return_filters=True, callback=lambda y: seen.append(y.shape))
    assert seen == [X.shape[:2] + (K,)] * 2
    floor = orc.rel_err(g["W_c64_laplace_20"], g["W_c128_laplace_20"])
    assert orc.rel_err(W, g["W_c128_laplace_20"]) < max(TOL, 1.5 * floor)
    assert oa.last_solver_info()["resident_launches"] >= 2
    # complex128 input (`precise`) at 4 channels runs in the kernel too
    Y128, W128 = oa.overiva(g["X"].astype(np.complex128), n_src=K, n_iter=20, proj_back=False, return_filters=True)
    assert oa.last_solver_info()["precision"] == "precise" and oa.last_solver_info()["resident_launches"] >= 1
    assert orc.rel_err(W128, g["W_c128_laplace_20"]) < TOL * max(1.0, float(g["amp_laplace_20"]) / 10.0)
