"""The two-lanes-per-(bin, frame) covariance kernels of 8-channel plans (reference overiva.py:179 and :87): the float64
form of the `precise` arithmetic (csrc/kernels_cov_pair64.hip, which replaced the fp64 matrix-core kernel) and the
float32 form with four sources per pass (csrc/kernels_cov_pair32.hip, three or more sources) -- against the oracle on
ragged shapes, and at full size through invariants that need no oracle.  The 4-channel `precise` pass (the plain kernel with
float64 accumulators) rides along."""
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


def _covariances(oa, T, F, M, K, mode, splits, seed=4):
    X = orc.synth_mixture(T, F, M, K, seed=seed)
    rinv = np.random.default_rng(5).gamma(2.0, 1.0, (T, K)).astype(np.float32)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision(mode)
        if splits:
            p.set_cov_splits(min(splits, max(1, T // 8)))
        p.set_x(X)
        p.covariance()
        Cx = p.get_cx()
        p.t_set_rinv(rinv)
        p.t_run_weighted_cov()
        V = p.t_get_v(np.complex128)
    # precise: the device divides 1 by the float32 reciprocal it is handed -- compare with exactly those weights
    w = 1.0 / (np.float32(1) / rinv).astype(np.float64) if mode == "precise" else rinv.astype(np.float64)
    return (orc.rel_err(V, orc.weighted_cov_all(X, w)), orc.rel_err(Cx, orc.input_covariance(X.astype(np.complex128))),
            bool(np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))))


@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("shape", [(144, 24, 8, 8), (150, 37, 8, 3), (61, 5, 8, 4), (9, 70, 8, 5), (203, 33, 8, 2), (77, 64, 8, 1),
                                   (130, 21, 4, 2), (40, 100, 4, 4)], ids=lambda s: "x".join(str(v) for v in s))
def test_precise_covariances_against_oracle(oa, shape, splits):
    """float64 sums of exact products: 1e-12 of the oracle's float64 result (bins that are no multiple of 32, frames that are
    no multiple of the 8-frame step or fewer than one step, odd source counts, several passes)"""
    eV, eC, herm = _covariances(oa, *shape, "precise", splits)
    print(f"\n[pair64] {shape} splits={splits}: V {eV:.1e} Cx {eC:.1e}")
    assert eV < 1e-12 and eC < 1e-12 and herm


@pytest.mark.parametrize("mode", ["fast", "mixed"])
@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("shape", [(144, 24, 8, 8), (150, 37, 8, 3), (61, 5, 8, 4), (9, 70, 8, 5), (300, 65, 8, 7)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_four_sources_per_pass_against_oracle(oa, shape, splits, mode):
    eV, eC, herm = _covariances(oa, *shape, mode, splits)
    print(f"\n[pair32] {shape} {mode} splits={splits}: V {eV:.1e} Cx {eC:.1e}")
    assert eV < 2e-7 and eC < 2e-7 and herm


def test_frame_splits_of_the_four_source_kernel(oa):
    """`mixed` (float64 per-bin algebra behind the pass) takes at least 8 frame splits, `fast` one round of workgroups;
    changing the arithmetic of an existing plan re-chooses"""
    with oa.Plan(4000, 2048, 8, 4, "laplace") as p:
        p.set_precision("fast")
        assert p.cov_splits() == 8          # 64 bin groups x 8 splits = 2 workgroups per CU
        p.set_precision("mixed")
        assert p.cov_splits() == 8
    # a short frame axis: the float32 chain of a lane, T / (4 splits) frames, stays <= 64 (what 4 splits give just below 1024
    # frames) -- round 5: not 4 splits whatever T, which at the reference's 235 frames cost 5 us of a 21 us pass
    with oa.Plan(200, 512, 8, 3, "laplace") as p:
        p.set_precision("fast")
        few = p.cov_splits()
        p.set_precision("mixed")
        assert p.cov_splits() == few and 200 / (4 * few) <= 64
    with oa.Plan(900, 512, 8, 3, "laplace") as p:
        p.set_precision("mixed")
        assert p.cov_splits() >= 4 and 900 / (4 * p.cov_splits()) <= 64
    with oa.Plan(235, 2049, 8, 4, "laplace") as p:          # the reference's own sweep shape
        p.set_precision("mixed")
        assert p.cov_splits() == 3


@pytest.mark.parametrize("case", [(4, "mixed"), (4, "fast"), (2, "precise"), (4, "precise")], ids=lambda c: f"{c[0]}src-{c[1]}")
def test_8_channels_full_size_properties(oa, case):
    """2048 bins x 4000 frames x 8 channels at FULL size on the split kernels (4 sources: four per pass; `precise`: float64):
    invariants that need no oracle, plus the covariances of three bins against the oracle"""
    K, mode = case
    T, F, M = 4000, 2048, 8
    X = orc.synth_iid(T, F, M, seed=2)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision(mode)
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.iterate(2)
        rinv, wscale = p.t_get_rinv()
        What = p.t_get_what(np.complex128)
        V = p.t_get_v(np.complex128)
        W = p.get_w()
    assert np.all(np.isfinite(W)) and W.shape == (F, M, K)
    assert abs(np.mean(1.0 / rinv.astype(np.float64), axis=0) - 1.0).max() < 1e-5       # overiva.py:158-159
    s = K - 1
    w = What[:, :, s]
    q = np.einsum("fc,fcd,fd->f", np.conj(w), V[s], w)                                  # overiva.py:185-186
    assert np.abs(q - 1.0).max() < (1e-4 if mode == "fast" else 1e-6 if mode == "mixed" else 1e-9)
    assert np.array_equal(V, np.conj(np.swapaxes(V, -1, -2)))
    for f in (0, 1023, 2047):
        ref = orc.weighted_cov_all(X[:, f:f + 1, :], rinv.astype(np.float64))[:, 0]
        assert orc.rel_err(V[:, f], ref) < 1e-6        # (the weights travel as float32 in every mode)
