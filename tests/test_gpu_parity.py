"""Parity of the HIP path (through the C ABI, ctypes) against the oracle and the committed golden
vectors of the real reference.  Needs an MI355X: run with ``-m gpu``.

Tolerances.  BASELINE.json's north_star asks for 1e-5 relative (fp32) on demixed Y and final W for
identical STFT input.  The distance is ||a-b||_F / ||b||_F against the reference's complex128 result
(the reference's own complex64 run sits 7e-7 .. 3e-5 from it).  Where the
reference itself amplifies rounding (fixture key ``amp_*`` = measured amplification of a relative
input perturbation, see tests/golden/make_golden.py) the bound is scaled by that amplification; where
it is chaotic (amp > 1e3) nothing is compared (conftest.chaotic).
"""
import numpy as np
import pytest

from conftest import chaotic, golden_files, golden_ids
from oracle import overiva_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5          # the north_star bound
TOL_KERNEL = 3e-6   # single-kernel bound (one fp32 pass, no iteration feedback)
TOL_KERNEL_F64 = 1e-12   # float64 accumulation of exact float32 products


def _amp(g, model, n_iter):
    return max(1.0, float(g.get(f"amp_{model}_{n_iter}", 1.0)))


def _c64_floor(fn_ref64, ref128):
    """distance of the reference algorithm's OWN complex64 arithmetic (oracle, reference-faithful
    mode) from its complex128 result on the same input: no float32 implementation can be expected
    to sit closer to the complex128 result than a small multiple of this."""
    try:
        with np.errstate(all="ignore"):
            e = orc.rel_err(fn_ref64(), ref128)
    except np.linalg.LinAlgError:
        return np.inf
    return e if np.isfinite(e) else np.inf


@pytest.fixture(scope="module")
def oa():
    import overiva_amd
    from overiva_amd import _lib

    _lib.load()
    return overiva_amd


def _plan(oa, X, K, model="laplace", mode="fast"):
    T, F, M = X.shape
    p = oa.Plan(T, F, M, K, model)
    p.set_precision(mode)
    p.set_x(X)
    p.covariance()
    return p


# --------------------------------------------------------------------------------------------
# per-kernel parity
# --------------------------------------------------------------------------------------------
def test_input_covariance(oa, golden):
    X, K = golden["X"], int(golden["K"])
    with _plan(oa, X, K) as p:
        Cx = p.get_cx()
    ref = orc.input_covariance(X.astype(np.complex128))
    assert orc.rel_err(Cx, ref) < TOL_KERNEL


@pytest.mark.parametrize("mode", ["fast", "precise"])
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_weighted_covariance(oa, golden, splits, mode):
    X, K = golden["X"], int(golden["K"])
    T = X.shape[0]
    rng = np.random.default_rng(5)
    rinv = rng.gamma(2.0, 1.0, (T, K)).astype(np.float32)
    with _plan(oa, X, K, mode=mode) as p:
        if splits:
            p.set_cov_splits(splits)
        p.t_set_rinv(rinv)
        p.t_run_weighted_cov()
        V = p.t_get_v(np.complex128)
    # the device divides 1 by the float32 reciprocal it is handed: compare with exactly those weights
    w = 1.0 / (np.float32(1) / rinv).astype(np.float64) if mode == "precise" else rinv.astype(np.float64)
    ref = orc.weighted_cov_all(X, w)
    assert V.shape == ref.shape
    e = orc.rel_err(V, ref)
    print(f"\n[parity] {golden['_id']} weighted covariance {mode} splits={splits}: {e:.2e}")
    # (more than 8 channels: the planar kernel takes its weights from a float32 table, 6e-8 each)
    tol64 = TOL_KERNEL_F64 if X.shape[2] <= 8 else 2e-7
    assert e < (tol64 if mode == "precise" else TOL_KERNEL)
    # Hermitian by construction
    assert np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))


def test_demix_power(oa, golden):
    X, K = golden["X"], int(golden["K"])
    T, F, M = X.shape
    rng = np.random.default_rng(6)
    What = (rng.standard_normal((F, M, M)) + 1j * rng.standard_normal((F, M, M))).astype(np.complex64)
    with _plan(oa, X, K) as p:
        p.t_set_what(What)
        pw = p.t_run_power()
    ref = orc.demix_power(X, What[:, :, :K])
    assert orc.rel_err(pw, ref) < TOL_KERNEL


@pytest.mark.parametrize("rows", [False, True], ids=["lane-per-element", "lane-per-row"])
@pytest.mark.parametrize("fp64", [False, True], ids=["f32", "f64"])
@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_ip_update(oa, golden, model, fp64, rows):
    """weighted covariance + per-bin chain (IP1 solve, normalisation, J) from the reference's own
    traced state at overiva.py:181 (W_hat after gamma scaling, r_inv) -> W_hat after the epoch."""
    if f"im_{model}_e0_s0_V" not in golden:
        pytest.skip("no traced intermediates in this fixture")
    X, K = golden["X"], int(golden["K"])
    T, F, M = X.shape
    for e in (0, 1):
        rinv = golden[f"im_{model}_e{e}_s0_rinv"]
        W_in = golden[f"im_{model}_e{e}_s0_What"]
        with _plan(oa, X, K, model) as p:
            p.set_precision(1 if fp64 else 0, row_layout=rows)
            p.set_w(None)                  # marks the plan ready; state is overwritten next
            p.t_set_what(W_in)
            p.t_set_rinv(rinv)
            p.t_run_weighted_cov()
            p.t_run_update()
            W_out = p.t_get_what()
        V = orc.weighted_cov_all(X, rinv)
        ref = orc.ip_update_bin(W_in, V, orc.input_covariance(X.astype(np.complex128)), K)
        assert orc.rel_err(W_out, ref) < 2e-5, (e, orc.rel_err(W_out, ref))


def test_j_initialisation(oa, golden):
    X, K = golden["X"], int(golden["K"])
    T, F, M = X.shape
    with _plan(oa, X, K) as p:
        p.set_w(None)
        What = p.t_get_what()
        Wg = p.get_w()
    ref = orc.init_demixing(orc.input_covariance(X.astype(np.complex128)), K)
    assert orc.rel_err(What, ref) < 1e-5
    assert orc.rel_err(Wg, ref[:, :, :K]) == 0.0


def test_activation(oa, golden):
    """one iteration, then r_inv and wscale against the oracle computed from the same start"""
    X, K = golden["X"], int(golden["K"])
    T, F, M = X.shape
    for model in ("laplace", "gauss"):
        with _plan(oa, X, K, model) as p:
            p.set_w(None)
            W0 = p.t_get_what()
            p.iterate(1)
            rinv, wscale = p.t_get_rinv()
        pw = orc.demix_power(X, W0[:, :, :K])
        ref_rinv, ref_ws = orc.finalize_activation(pw, F, model)
        assert orc.rel_err(rinv, ref_rinv) < TOL_KERNEL
        assert orc.rel_err(wscale, ref_ws) < TOL_KERNEL
        assert abs(np.mean(1.0 / rinv.astype(np.float64), axis=0) - 1.0).max() < 1e-5


# --------------------------------------------------------------------------------------------
# end to end: overiva() against the reference's golden outputs
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["laplace", "gauss"])
@pytest.mark.parametrize("n_iter", [0, 1, 2, 5, 20])
@pytest.mark.parametrize("dt", ["c64", "c128"])
def test_overiva_matches_reference(oa, golden, model, n_iter, dt):
    if chaotic(golden, model, n_iter):
        pytest.skip("reference is ill-conditioned here (see conftest.chaotic)")
    X, K = golden["X"], int(golden["K"])
    Xin = X if dt == "c64" else X.astype(np.complex128)
    Y, W = oa.overiva(Xin, n_src=K, n_iter=n_iter, proj_back=False, model=model, return_filters=True)
    assert Y.dtype == Xin.dtype and W.dtype == Xin.dtype
    assert Y.shape == (X.shape[0], X.shape[1], K) and W.shape == (X.shape[1], X.shape[2], K)
    assert W.flags["C_CONTIGUOUS"]
    refW = golden[f"W_c128_{model}_{n_iter}"]
    eW = orc.rel_err(W, refW)
    # floor: the real reference's own complex64 run on this input (stored next to its complex128 run)
    k64 = f"W_c64_{model}_{n_iter}"
    floor = orc.rel_err(golden[k64], refW) if k64 in golden else 0.0
    # (a float32 implementation lands within a small multiple of that floor, not below it: on the 16-channel
    # determined mixture with 64 frames the last-bit rounding of a reciprocal moves the result by 3x)
    bound = max(TOL * _amp(golden, model, n_iter), 5 * floor)
    print(f"\n[parity] {golden['_id']} {model} n_iter={n_iter} {dt}: W err {eW:.2e} (bound {bound:.1e})")
    assert eW < bound
    if n_iter == 20:
        eY = orc.rel_err(Y, golden[f"Y_c128_{model}_20"])
        print(f"[parity] {golden['_id']} {model} n_iter=20 {dt}: Y err {eY:.2e}")
        assert eY < bound


@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_proj_back_and_callback(oa, golden, model):
    if chaotic(golden, model, 12):
        pytest.skip("reference is ill-conditioned here (see conftest.chaotic)")
    X, K = golden["X"], int(golden["K"])
    floor = _c64_floor(lambda: orc.overiva_faithful(X, n_src=K, n_iter=12, proj_back=True, model=model),
                       golden[f"Ypb_c128_{model}_12"])
    if floor > 1e-3:
        pytest.skip(f"the reference's own complex64 run is {floor:.1e} away from its complex128 run here")
    got = []
    Y = oa.overiva(X.astype(np.complex128), n_src=K, n_iter=12, proj_back=True, model=model,
                   callback=lambda y: got.append(np.array(y)))
    bound = max(TOL * _amp(golden, model, 12), 10 * floor)
    print(f"\n[parity] {golden['_id']} {model} proj_back 12 its: Y err "
          f"{orc.rel_err(Y, golden[f'Ypb_c128_{model}_12']):.2e} (reference c64 floor {floor:.1e}, bound {bound:.1e})")
    assert orc.rel_err(Y, golden[f"Ypb_c128_{model}_12"]) < bound
    assert len(got) == 2 and got[0].shape == Y.shape and got[0].dtype == np.complex128
    if model == "laplace":
        assert orc.rel_err(got[0], golden["cb0_c128_laplace"]) < bound
        assert orc.rel_err(got[1], golden["cb10_c128_laplace"]) < bound


def test_warm_start_default_nsrc_eig(oa, golden):
    X, K = golden["X"], int(golden["K"])
    X128 = X.astype(np.complex128)
    _, W = oa.overiva(X128, n_src=K, n_iter=3, proj_back=False, W0=golden["W0"], return_filters=True)
    floor = _c64_floor(lambda: orc.overiva_faithful(X, n_src=K, n_iter=3, proj_back=False, W0=golden["W0"],
                                                    return_filters=True)[1], golden["W_w0_c128_laplace_3"])
    assert orc.rel_err(W, golden["W_w0_c128_laplace_3"]) < max(TOL, 5 * floor)
    _, W = oa.overiva(X128, n_iter=2, proj_back=False, return_filters=True)
    assert W.shape == golden["W_det_c128_laplace_2"].shape
    floor = _c64_floor(lambda: orc.overiva_faithful(X, n_iter=2, proj_back=False, return_filters=True)[1],
                       golden["W_det_c128_laplace_2"])
    print(f"\n[parity] {golden['_id']} determined 2 its: W err "
          f"{orc.rel_err(W, golden['W_det_c128_laplace_2']):.2e} (reference c64 floor {floor:.1e})")
    assert orc.rel_err(W, golden["W_det_c128_laplace_2"]) < max(TOL, 5 * floor)
    Y = oa.overiva(X128, n_src=K, n_iter=3, proj_back=False, init_eig=True)
    # eigenvector phase is LAPACK's choice: compare magnitudes
    assert orc.rel_err(np.abs(Y), np.abs(golden["Y_eig_c128_laplace_3"])) < 1e-4


def test_auxiva_pca(oa, golden):
    X, K = golden["X"], int(golden["K"])
    Y = oa.auxiva_pca(X.astype(np.complex128), n_src=K, n_iter=5, proj_back=True, model="laplace")
    assert Y.shape == golden["Ypca_c128_laplace_5"].shape and Y.dtype == np.complex128
    assert orc.rel_err(Y, golden["Ypca_c128_laplace_5"]) < 1e-4
    with pytest.raises(KeyError):
        oa.auxiva_pca(X.astype(np.complex128), n_src=K, n_iter=1)


@pytest.mark.parametrize("shape", [(96, 5, 11, 3), (80, 3, 9, 9), (72, 6, 13, 1), (64, 4, 16, 5), (50, 7, 7, 7), (90, 19, 5, 2),
                                   (90, 21, 8, 3), (64, 9, 8, 4), (70, 5, 8, 5), (100, 12, 8, 8), (60, 7, 7, 4)])
def test_odd_shapes_against_oracle(oa, shape):
    """channel counts without a golden fixture (incl. the matrix-core covariance path, 9..16 channels)"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=sum(shape))
    for model in ("laplace", "gauss"):
        Y, W = oa.overiva(X, n_src=K, n_iter=4, proj_back=True, model=model, return_filters=True)
        Yr, Wr = orc.overiva_staged(X, n_src=K, n_iter=4, proj_back=True, model=model, return_filters=True)
        floor = _c64_floor(lambda: orc.overiva_faithful(X, n_src=K, n_iter=4, proj_back=True, model=model,
                                                        return_filters=True)[1], Wr)
        eW, eY = orc.rel_err(W, Wr), orc.rel_err(Y, Yr)
        print(f"\n[parity] T{T} F{F} M{M} K{K} {model} 4 its: W err {eW:.2e} Y err {eY:.2e} (reference c64 floor {floor:.1e})")
        assert eW < max(TOL, 5 * floor) and eY < max(TOL, 5 * floor)


@pytest.mark.parametrize("shape", [(5, 1, 1, 1), (17, 3, 2, 1), (33, 70, 3, 3), (5000, 2, 4, 2), (16, 16, 8, 8),
                                   (2, 5, 2, 2), (1000, 1, 6, 2)])
def test_ragged_and_extreme_shapes(oa, shape):
    """frames not a multiple of the 16-frame step, fewer bins than a 16-bin wave, single bin / channel,
    very long and very short frame axes"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=7 + sum(shape))
    Y, W = oa.overiva(X, n_src=K, n_iter=3, proj_back=False, return_filters=True)
    Yr, Wr = orc.overiva_staged(X, n_src=K, n_iter=3, proj_back=False, return_filters=True)
    if not np.all(np.isfinite(Wr)):
        pytest.skip("degenerate for the algorithm itself (oracle non-finite)")
    floor = _c64_floor(lambda: orc.overiva_faithful(X, n_src=K, n_iter=3, proj_back=False, return_filters=True)[1], Wr)
    assert orc.rel_err(W, Wr) < max(TOL, 5 * floor) and orc.rel_err(Y, Yr) < max(TOL, 5 * floor)


def test_config1_shape_against_oracle(oa):
    """BASELINE.json configs[0] shape: 2049 bins x ~160 frames x 4 mics / 2 src, complex128 in, 20 iterations,
    proj_back and the callback every 10 epochs (overiva_oneshot.py -a overiva -m 4 -s 2 -n 20)"""
    X = orc.synth_mixture(160, 2049, 4, 2, seed=3).astype(np.complex128)
    got, ref_got = [], []
    Y = oa.overiva(X, n_src=2, n_iter=20, proj_back=True, callback=lambda y: got.append(y.copy()))
    Yr = orc.overiva_staged(X, n_src=2, n_iter=20, proj_back=True, callback=lambda y: ref_got.append(y.copy()))
    floor = _c64_floor(lambda: orc.overiva_faithful(X.astype(np.complex64), n_src=2, n_iter=20, proj_back=True), Yr)
    e = orc.rel_err(Y, Yr)
    print(f"\n[parity] cfg1 shape mixture 20 its proj_back: Y err {e:.2e} (reference c64 floor {floor:.1e})")
    assert Y.dtype == np.complex128 and len(got) == len(ref_got) == 2
    assert e < max(TOL, 5 * floor)
    assert orc.rel_err(got[1], ref_got[1]) < max(TOL, 5 * floor)


def test_errors(oa):
    X = orc.synth_iid(32, 4, 3, seed=1)
    with pytest.raises(ValueError):
        oa.overiva(X, n_src=2, model="cauchy")
    with pytest.raises(ValueError):
        oa.overiva(X, n_src=4)
    with pytest.raises(TypeError):
        oa.overiva(X.real, n_src=2)
    # a rank-deficient input makes W_hat^H V singular: numpy raises LinAlgError in the reference
    Xz = np.zeros((32, 4, 3), np.complex64)
    with pytest.raises(np.linalg.LinAlgError):
        oa.overiva(Xz, n_src=2, n_iter=1, proj_back=False)


def test_input_not_mutated_and_noncontiguous(oa):
    X = orc.synth_iid(64, 9, 4, seed=2)
    Xt = np.asfortranarray(X)            # non C-contiguous view of the same values
    keep = X.copy()
    Y1 = oa.overiva(X, n_src=2, n_iter=3, proj_back=True)
    Y2 = oa.overiva(Xt, n_src=2, n_iter=3, proj_back=True)
    assert np.array_equal(X, keep)
    assert np.array_equal(Y1, Y2)        # deterministic: same input, bitwise same output


def test_graph_replay_equals_eager(oa):
    X = orc.synth_iid(128, 40, 4, seed=3)
    outs = []
    for graph in (False, True):
        with _plan(oa, X, 2) as p:
            p.use_graph(graph)
            p.set_w(None)
            p.iterate(3)
            p.iterate(2)
            p.iterate(19)          # 2 batches of 8 iterations + 3 single replays
            outs.append(p.get_w())
    assert np.array_equal(outs[0], outs[1])


# --------------------------------------------------------------------------------------------
# BASELINE configs: cfg2 against the oracle, headline size through size-independent properties
# --------------------------------------------------------------------------------------------
def test_cfg2_against_oracle(oa):
    """513 bins x 1000 frames x 4 mics / 2 src, laplace (BASELINE.json configs[1])"""
    X = orc.synth_iid(1000, 513, 4, seed=0)
    Y, W = oa.overiva(X, n_src=2, n_iter=10, proj_back=False, return_filters=True)
    Yr, Wr = orc.overiva_staged(X, n_src=2, n_iter=10, proj_back=False, return_filters=True)
    eW, eY = orc.rel_err(W, Wr), orc.rel_err(Y, Yr)
    print(f"\n[parity] cfg2 iid 10 its: W err {eW:.2e}  Y err {eY:.2e}")
    assert eW < TOL and eY < TOL


def test_cfg2_mixture_against_oracle(oa):
    X = orc.synth_mixture(1000, 513, 4, 2, seed=1)
    Y, W = oa.overiva(X, n_src=2, n_iter=10, proj_back=False, return_filters=True)
    Yr, Wr = orc.overiva_staged(X, n_src=2, n_iter=10, proj_back=False, return_filters=True)
    eW, eY = orc.rel_err(W, Wr), orc.rel_err(Y, Yr)
    # floor: the reference's own complex64 arithmetic on this input
    _, Wf = orc.overiva_faithful(X, n_src=2, n_iter=10, proj_back=False, return_filters=True)
    floor = orc.rel_err(Wf, Wr)
    print(f"\n[parity] cfg2 mixture 10 its: W err {eW:.2e}  Y err {eY:.2e}  (reference c64 floor {floor:.2e})")
    assert eY < 5e-5 and eW < max(5e-5, 3 * floor)


def test_headline_size_against_oracle(oa):
    """2048 x 4000 x 8 / 2 (BASELINE.json configs[2]) end to end against the oracle's reference-faithful form
    (the reference's own arithmetic: complex64 data, float64 activations) for a few iterations."""
    T, F, M, K = 4000, 2048, 8, 2
    X = orc.synth_iid(T, F, M, seed=0)
    Y, W = oa.overiva(X, n_src=K, n_iter=3, proj_back=False, return_filters=True)
    Yr, Wr = orc.overiva_faithful(X, n_src=K, n_iter=3, proj_back=False, return_filters=True)
    eW, eY = orc.rel_err(W, Wr), orc.rel_err(Y, Yr)
    print(f"\n[parity] headline shape iid 3 its vs reference-faithful c64: W err {eW:.2e}  Y err {eY:.2e}")
    assert eW < TOL and eY < TOL


def test_headline_size_properties(oa):
    """2048 bins x 4000 frames x 8 mics / 2 src (BASELINE.json configs[2]): invariants that need no
    oracle, plus a spot check of a few bins against the oracle."""
    T, F, M, K = 4000, 2048, 8, 2
    X = orc.synth_iid(T, F, M, seed=0)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.iterate(3)
        rinv, wscale = p.t_get_rinv()
        What = p.t_get_what().astype(np.complex128)
        V = p.t_get_v().astype(np.complex128)          # covariances of the last iteration
        Cx = p.get_cx().astype(np.complex128)
        W = p.get_w()
    assert np.all(np.isfinite(W))
    # mean_t r = 1 before the eps floor (overiva.py:158-159)
    assert abs(np.mean(1.0 / rinv.astype(np.float64), axis=0) - 1.0).max() < 1e-5
    # w_s^H V_s w_s = 1 for the last-updated source (overiva.py:185-186)
    s = K - 1
    w = What[:, :, s]
    q = np.einsum("fc,fcd,fd->f", np.conj(w), V[s], w)
    assert np.abs(q - 1.0).max() < 1e-4
    # orthogonality constraint W^H Cx [J; -I] = 0 (overiva.py:96-98, :123)
    R = np.conj(np.swapaxes(What[:, :, :K], 1, 2)) @ Cx @ What[:, :, K:]
    scale = np.abs(np.conj(np.swapaxes(What[:, :, :K], 1, 2)) @ Cx).max()
    assert np.abs(R).max() < 1e-4 * scale
    # spot check: covariances of 3 bins against the oracle, given the device's own r_inv
    for f in (0, 1023, 2047):
        ref = orc.weighted_cov_all(X[:, f:f + 1, :], rinv.astype(np.float64))[:, 0]
        assert orc.rel_err(V[:, f], ref) < TOL_KERNEL
    # and the whole run against the oracle on a bin subset is impossible (r couples all bins); instead
    # the activation itself: r from the device's W before the last iteration is covered by
    # test_activation at small sizes; here only its normalisation is checked (above).


def test_plain_c_program_runs(oa, tmp_path):
    """examples/c_abi_demo.c: the C ABI driven from plain C on the GPU"""
    import os
    import subprocess

    from conftest import REPO

    exe = tmp_path / "c_abi_demo"
    pkg = os.path.join(REPO, "overiva_amd")
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", "c_abi_demo.c"),
                        "-L", pkg, "-loveriva_hip", f"-Wl,-rpath,{pkg}", "-lm", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "output energy" in run.stdout


def test_two_plans_are_independent_and_no_leak(oa):
    """distinct plans may be alive and interleaved; repeated create/destroy does not accumulate device memory"""
    import torch

    Xa = orc.synth_iid(128, 40, 4, seed=11)
    Xb = orc.synth_iid(96, 33, 3, seed=12)
    Ya = oa.overiva(Xa, n_src=2, n_iter=4, proj_back=False)
    Yb = oa.overiva(Xb, n_src=1, n_iter=4, proj_back=False)
    pa = oa.Plan(128, 40, 4, 2)
    pb = oa.Plan(96, 33, 3, 1)
    pa.set_x(Xa); pb.set_x(Xb)
    pa.covariance(); pb.covariance()
    pa.set_w(None); pb.set_w(None)
    for _ in range(4):                      # interleaved iterations on two streams
        pa.iterate(1)
        pb.iterate(1)
    assert np.array_equal(pa.demix(False), Ya) and np.array_equal(pb.demix(False), Yb)
    pa.close(); pb.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        with oa.Plan(512, 256, 8, 2) as p:
            p.set_x(orc.synth_iid(512, 256, 8, seed=1))
            p.covariance(); p.set_w(None); p.iterate(2); p.demix(True)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20
