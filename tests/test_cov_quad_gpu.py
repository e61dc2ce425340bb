"""The vector-ALU covariance kernels of 10 / 12 / 14 / 16-channel plans (reference overiva.py:179 and :87): four lanes per
(bin, frame) for up to 4 sources (csrc/kernels_cov_quad.hip), 32 lanes per (bin, frame) and every source in one pass for
5..16 (csrc/kernels_cov_half16.hip, float32 and float64 sums) -- against the oracle on ragged shapes, against the matrix-core kernel they replace,
the rules that select them, the default arithmetic of these shapes, and the full-size geometry (2048 bins x 4000 frames x
16 channels / 2 sources; BASELINE configs[4] is tests/test_gpu_parity.py::test_cfg5_full_size_properties) through invariants
that need no oracle."""
import numpy as np
import pytest

from oracle import overiva_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oa():
    import overiva_amd
    from overiva_amd import _lib

    _lib.load()
    return overiva_amd


SHAPES = [(163, 19, 16, 2, "fast"), (150, 18, 10, 2, "fast"), (141, 17, 14, 1, "fast"), (160, 16, 12, 3, "mixed"),
          (200, 33, 16, 4, "mixed"), (61, 5, 16, 2, "mixed"), (9, 3, 10, 1, "fast"), (35, 40, 12, 2, "fast"),
          # odd channel counts: the same kernels on the copy of X padded by a zero channel
          (157, 21, 15, 2, "fast"), (140, 18, 9, 1, "mixed"), (150, 9, 13, 4, "mixed"), (131, 7, 11, 3, "mixed")]


@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(str(v) for v in s))
def test_quad_covariances_against_oracle(oa, shape, splits):
    """V_k (overiva.py:179, all sources) and Cx (overiva.py:87): bins that are no multiple of 16, frames that are no
    multiple of the 8-frame step or fewer than one step, 10 / 12 / 14 channels (entries of channels past M are dropped)"""
    T, F, M, K, mode = shape
    X = orc.synth_mixture(T, F, M, K, seed=4)
    rinv = np.random.default_rng(5).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision(mode)
        assert p.set_cov_quad(True)
        if splits:
            p.set_cov_splits(min(splits, max(1, T // 8)))
        p.set_x(X)
        p.covariance()
        Cx = p.get_cx()
        p.t_set_rinv(rinv)
        p.t_run_weighted_cov()
        V = p.t_get_v(np.complex128)
    eC = orc.rel_err(Cx, orc.input_covariance(X.astype(np.complex128)))
    eV = orc.rel_err(V, orc.weighted_cov_all(X, rinv.astype(np.float64)))
    print(f"\n[quad] {shape} splits={splits}: V {eV:.1e} Cx {eC:.1e}")
    assert eV < 2e-7 and eC < 2e-7
    assert np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))


MANY = [(64, 4, 16, 16, "fast"), (163, 19, 16, 9, "mixed"), (150, 18, 10, 5, "fast"), (141, 17, 14, 8, "mixed"),
        (33, 3, 12, 12, "fast"), (9, 5, 16, 7, "fast"), (200, 7, 16, 12, "mixed"), (17, 1, 14, 14, "mixed"),
        (150, 11, 15, 15, "mixed"), (64, 6, 9, 5, "fast"), (99, 3, 13, 9, "fast"), (40, 1, 11, 11, "mixed")]


@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("shape", MANY, ids=lambda s: "x".join(str(v) for v in s))
def test_many_source_covariances_against_oracle(oa, shape, splits):
    """5..16 sources in one pass: an odd number of bins (the last workgroup holds one), a single bin, frames that are no
    multiple of the 16-frame stage or fewer than one stage, 10 / 12 / 14 channels, source counts that are no multiple of
    the instantiated 8 / 12 / 16"""
    T, F, M, K, mode = shape
    X = orc.synth_mixture(T, F, M, K, seed=4)
    rinv = np.random.default_rng(5).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision(mode)
        assert p.set_cov_quad(True)
        if splits:
            p.set_cov_splits(min(splits, max(1, T // 16)))
        p.set_x(X)
        p.covariance()
        Cx = p.get_cx()
        p.t_set_rinv(rinv)
        p.t_run_weighted_cov()
        V = p.t_get_v(np.complex128)
    eC = orc.rel_err(Cx, orc.input_covariance(X.astype(np.complex128)))
    eV = orc.rel_err(V, orc.weighted_cov_all(X, rinv.astype(np.float64)))
    print(f"\n[half16] {shape} splits={splits}: V {eV:.1e} Cx {eC:.1e}")
    assert eV < 2e-7 and eC < 2e-7
    assert np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))


@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("shape", MANY + [s for s in SHAPES if s[3] >= 3], ids=lambda s: "x".join(str(v) for v in s[:4]))
def test_covariances_in_float64_against_oracle(oa, shape, splits):
    """`precise` with 3..16 sources: the lanes of the many-source kernel, float64 sums of exact products, 4 or 8 sources per
    pass (source counts below, at and above one and two passes); Cx and one or two sources stay on the matrix-core kernel"""
    T, F, M, K, _ = shape
    X = orc.synth_mixture(T, F, M, K, seed=4)
    rinv = np.random.default_rng(5).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision("precise")
        assert p.set_cov_quad(True)
        if splits:
            p.set_cov_splits(min(splits, max(1, T // 16)))
        p.set_x(X)
        p.covariance()
        Cx = p.get_cx()
        p.t_set_rinv(rinv)
        p.t_run_weighted_cov()
        V = p.t_get_v(np.complex128)
        p.set_cov_quad(False)
        p.t_run_weighted_cov()
        Vm = p.t_get_v(np.complex128)
    eC = orc.rel_err(Cx, orc.input_covariance(X.astype(np.complex128)))
    w = 1.0 / (np.float32(1) / rinv).astype(np.float64)      # the hook stores r = 1 / rinv in float32; the weight is 1 / r in float64
    eV = orc.rel_err(V, orc.weighted_cov_all(X, w))
    eM = orc.rel_err(V, Vm)                                    # (the matrix-core kernel takes its weights in float32)
    print(f"\n[half16 f64] {shape[:4]} splits={splits}: V {eV:.1e} Cx {eC:.1e} vs matrix-core {eM:.1e}")
    assert eV < 1e-12 and eC < 1e-12 and eM < 1e-7
    assert np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))


def test_selection_rules(oa):
    """one or two sources: every float32 mode; three or four: only with the float64 per-bin algebra (the matrix-core kernel is
    faster in `fast`); five and more: the 32-lanes-per-(bin, frame) kernel; `precise`: the float64 form of that kernel for
    three and more sources; 9 / 11 / 13 / 15 channels follow the rules of the next even count (the kernels then read a copy
    of X padded by one zero channel); the switch turns all of them off"""
    def active(M, K, mode, on=True):
        with oa.Plan(64, 20, M, K, "laplace") as p:
            p.set_precision(mode)
            return p.set_cov_quad(on)

    assert active(16, 2, "fast") and active(10, 1, "fast") and active(14, 2, "mixed")
    assert active(12, 3, "mixed") and active(16, 4, "mixed")
    assert not active(12, 3, "fast") and not active(16, 4, "fast")
    assert not active(8, 2, "fast") and not active(7, 7, "mixed")
    assert active(11, 2, "fast") and active(9, 1, "mixed") and active(13, 4, "mixed") and not active(13, 4, "fast")      # odd: as the next even count
    assert active(15, 15, "fast") and active(9, 5, "mixed") and active(15, 9, "precise") and active(13, 4, "precise")
    assert not active(15, 2, "precise") and not active(15, 2, "fast", on=False)
    assert active(16, 5, "mixed") and active(16, 16, "fast") and active(12, 12, "mixed")      # many sources: kernels_cov_half16.hip
    assert active(16, 5, "precise") and active(16, 16, "precise") and active(10, 3, "precise") and active(12, 4, "precise")
    assert not active(16, 2, "precise") and not active(16, 1, "precise")
    assert not active(16, 2, "fast", on=False)
    # the precision set AFTER the switch decides as well
    with oa.Plan(64, 20, 12, 3, "laplace") as p:
        p.set_precision("fast")
        assert not p.set_cov_quad(True)
        p.set_precision("mixed")
        assert p.set_cov_quad(True)
        p.set_precision("precise")
        assert p.set_cov_quad(True)             # (the float64 form of the many-source kernel)


@pytest.mark.parametrize("shape", [(163, 19, 16, 2), (150, 18, 10, 2), (141, 33, 14, 1), (128, 9, 16, 16), (120, 12, 12, 7),
                                   (157, 21, 15, 2), (140, 10, 9, 9), (133, 7, 13, 4)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_same_iteration_as_the_matrix_core_kernel(oa, shape):
    """5 iterations with either covariance kernel under the same float64 per-bin algebra: the two differ only in the
    rounding of the float32 partial sums"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=8)
    W = {}
    for quad in (True, False):
        with oa.Plan(T, F, M, K, "laplace") as p:
            p.set_precision("mixed")
            assert p.set_cov_quad(quad) == quad
            p.set_x(X)
            p.covariance()
            p.set_w(None)
            p.iterate(5)
            W[quad] = p.get_w(np.complex128)
    e = orc.rel_err(W[True], W[False])
    print(f"\n[quad] {shape}: W vector-ALU vs matrix-core {e:.1e}")
    assert e < (5e-6 if K <= 4 else 5e-5)       # (many sources: the determined problem amplifies the float32 partials' noise)


def test_default_arithmetic_of_10_to_16_channels(oa):
    """complex64 input: `mixed` (the vector-ALU covariance kernels; odd channel counts on the padded copy of X);
    complex128 input: `precise`"""
    rng = np.random.default_rng(0)
    for (M, K, dt, want) in ((16, 2, np.complex64, "mixed"), (12, 4, np.complex64, "mixed"), (16, 5, np.complex64, "mixed"), (16, 16, np.complex64, "mixed"), (13, 13, np.complex64, "mixed"),
                             (11, 2, np.complex64, "mixed"), (16, 2, np.complex128, "precise"), (15, 6, np.complex128, "precise")):
        X = (rng.standard_normal((40, 6, M)) + 1j * rng.standard_normal((40, 6, M))).astype(dt)
        Y = oa.overiva(X, n_src=K, n_iter=2, proj_back=False)
        assert Y.dtype == dt and oa.last_solver_info()["precision"] == want


@pytest.mark.parametrize("mode", ["mixed", "fast"])
def test_16_channels_2_sources_full_size_properties(oa, mode):
    """2048 bins x 4000 frames x 16 channels / 2 sources at FULL size -- the geometry the kernel was built for (128 bin
    groups x 8 or 4 frame splits, 2 workgroups per CU, 1.05 GB of X): invariants that need no oracle, plus the
    covariances of three bins against the oracle"""
    T, F, M, K = 4000, 2048, 16, 2
    X = orc.synth_iid(T, F, M, seed=2)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision(mode)
        assert p.set_cov_quad(True)
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.iterate(2)
        rinv, wscale = p.t_get_rinv()
        What = p.t_get_what(np.complex128)
        V = p.t_get_v(np.complex128)                       # covariances of the last iteration, (K, F, M, M)
        W = p.get_w()
        splits = p.cov_splits()
    assert splits == (8 if mode == "mixed" else 4)
    assert np.all(np.isfinite(W)) and W.shape == (F, M, K)
    assert abs(np.mean(1.0 / rinv.astype(np.float64), axis=0) - 1.0).max() < 1e-5       # overiva.py:158-159
    s = K - 1
    w = What[:, :, s]
    q = np.einsum("fc,fcd,fd->f", np.conj(w), V[s], w)                                  # overiva.py:185-186
    assert np.abs(q - 1.0).max() < (1e-4 if mode == "fast" else 1e-6)
    assert np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))
    for f in (0, 1023, 2047):
        ref = orc.weighted_cov_all(X[:, f:f + 1, :], rinv.astype(np.float64))[:, 0]
        assert orc.rel_err(V[:, f], ref) < 1e-6


@pytest.mark.parametrize("shape", [(400, 6, 16, 16), (333, 5, 16, 9), (250, 3, 12, 12), (200, 4, 14, 11), (180, 3, 15, 15), (170, 2, 11, 10)])
def test_sources_on_the_matrix_cores_against_the_vector_alu_kernel(oa, shape):
    """9..16 sources: the weighted sums of all sources as one small GEMM per bin on the fp32 matrix cores
    (csrc/kernels_cov_hmfma.hip), the Hermitian products formed along the cyclic diagonals of the matrix.  Same arithmetic
    class as the vector-ALU kernel -- float32 products, float32 chains (the fp32 matrix instruction is an fmaf chain in frame
    order), float64 sums across chains and splits -- with the frames dealt to the chains differently (eight chains per
    workgroup and half the splits), so the two kernels agree to a few float32 roundings and both sit within the single-pass
    bound of the oracle.  Ragged shapes: odd channel counts (padded copy of X), fewer than 16 channels / sources, frames that
    do not fill the last stage."""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=31)
    rinv = np.random.default_rng(5).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    V = {}
    for on in (True, False):
        with oa.Plan(T, F, M, K, "laplace") as p:
            p.set_precision("mixed")
            p.set_cov_hmfma(on)
            p.set_x(X)
            p.covariance()
            p.t_set_rinv(rinv)
            p.t_run_weighted_cov()
            V[on] = p.t_get_v(np.complex128)
    ref = orc.weighted_cov_all(X, rinv.astype(np.float64))
    assert orc.rel_err(V[True], V[False]) < 3e-7
    assert orc.rel_err(V[True], ref) < 3e-6 and orc.rel_err(V[False], ref) < 3e-6


@pytest.mark.parametrize("shape", [(33, 3, 16, 16), (97, 2, 16, 13), (160, 3, 16, 16), (223, 2, 15, 15), (129, 3, 12, 12), (96, 2, 14, 9), (2100, 2, 16, 16)])
def test_matrix_core_kernel_ring_tails_and_both_forms_of_its_loads(oa, shape, monkeypatch):
    """cov_hmfma_kernel, round 5: the ring of four stages is unrolled (a split of 1, 2, 3 stages and of 4 n + 1 .. 3 runs the
    tail code), the loads are buffer-form DMAs whose descriptor steps through the frames on the scalar unit and whose range check
    zeroes the frames past T (the flat form clamps addresses instead; $OIVA_HMFMA_FLAT=1), and at 16 channels the two parts of the
    pair distance 8 share one group of the matrix instruction.  The two forms of the loads give the SAME BITS; both sit within the
    single-pass bound of the oracle; one split and several."""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=37)
    rinv = np.random.default_rng(7).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    ref = orc.weighted_cov_all(X, rinv.astype(np.float64))
    for splits in (1, 0):
        V = {}
        # (flat, pk): the loads flat or buffer-form; at 16 channels the partners of the products from LDS with packed products
        # ($OIVA_HMFMA_PK, default) or by DPP rotations -- four forms of one arithmetic
        for flat, pk in (("0", "1"), ("1", "1"), ("0", "0"), ("1", "0")):
            monkeypatch.setenv("OIVA_HMFMA_FLAT", flat)
            monkeypatch.setenv("OIVA_HMFMA_PK", pk)
            with oa.Plan(T, F, M, K, "laplace") as p:
                p.set_precision("mixed")
                if splits:
                    p.set_cov_splits(splits)
                p.set_x(X)
                p.covariance()
                p.t_set_rinv(rinv)
                p.t_run_weighted_cov()
                V[flat + pk] = p.t_get_v(np.complex128)
        assert np.array_equal(V["01"], V["11"]) and np.array_equal(V["01"], V["00"]) and np.array_equal(V["01"], V["10"])
        assert orc.rel_err(V["01"], ref) < (2e-5 if splits == 1 and T > 2000 else 3e-6)
        assert np.array_equal(V["01"], np.conj(np.swapaxes(V["01"], -1, -2)))


@pytest.mark.parametrize("shape", [(400, 6, 16, 16), (333, 5, 16, 9), (180, 3, 15, 15), (70, 2, 16, 12)])
def test_sources_on_the_fp64_matrix_cores_against_the_vector_alu_kernel(oa, shape):
    """`precise`, 16 channels (15: padded copy of X), 9..16 sources: the same GEMM per bin on the fp64 matrix cores
    (cov_hmfma64_kernel) against the float64 vector-ALU kernel it replaces there and against the oracle: float64 sums of exact
    float64 products in both, different orders"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=33)
    rinv = np.random.default_rng(6).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    V = {}
    for on in (True, False):
        with oa.Plan(T, F, M, K, "laplace") as p:
            p.set_precision("precise")
            p.set_cov_hmfma(on)
            p.set_x(X)
            p.covariance()
            p.t_set_rinv(rinv)
            p.t_run_weighted_cov()
            V[on] = p.t_get_v(np.complex128)
    w = 1.0 / (np.float32(1) / rinv).astype(np.float64)
    ref = orc.weighted_cov_all(X, w)
    assert orc.rel_err(V[True], V[False]) < 1e-13
    assert orc.rel_err(V[True], ref) < 1e-12 and orc.rel_err(V[False], ref) < 1e-12
    assert np.array_equal(V[True], np.conj(np.swapaxes(V[True], -1, -2)))
