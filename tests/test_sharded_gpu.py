"""The bin-sharded product path on ONE MI355X: several plans with F_total > F on the same device, their partial
power buffers concatenated by a device copy (what the RCCL all-gather of overiva_amd/sharded.py delivers), against
the single-plan run.  Reference coupling: overiva.py:152-155 (r needs all bins) is the only exchange.

Checked: oiva_plan_power / oiva_plan_power_buffer / oiva_plan_update, the parts-per-rank layout with zero-padded
parts, the fixed part order of the sum, F_total in the gauss model's 1/F, plans running on a caller-provided
stream.  EQUAL shards on 64-bin batches must give the SAME BITS as the single plan (the sum over the parts is associated
in the same canonical blocks, csrc/kernels_misc.hip); other boundaries the same result to rounding (the zero parts that pad
unequal shards to one message size shift the blocks).  Needs an MI355X: run with ``-m gpu``.
"""
import numpy as np
import pytest

from oracle import overiva_oracle as orc

pytestmark = pytest.mark.gpu

T, F, M, K = 300, 448, 8, 2          # 7 batches of 64 bins


def _run_single(oa, X, model, mode, n_iter):
    F = X.shape[1]
    with oa.Plan(T, F, M, K, model) as p:
        p.set_precision(mode)
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.iterate(n_iter)
        return p.get_w(np.complex128), p.demix(True)


def _run_sharded(oa, X, model, mode, n_iter, bounds):
    import torch

    from overiva_amd.sharded import HipEngine

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    world = len(bounds) - 1
    F = X.shape[1]
    with torch.cuda.stream(stream):
        engines = [HipEngine(T, bounds[r + 1] - bounds[r], M, K, model, F, 0, precision=mode) for r in range(world)]
        assert all(e.stream.cuda_stream == stream.cuda_stream for e in engines)       # one stream orders everything
        ppr = max(e.power_parts(bounds[r + 1] - bounds[r]) for r, e in enumerate(engines))
        local = [e.exchange_buffer(ppr) for e in engines]
        gathered = engines[0].new_gather_buffer(world)
        for r, e in enumerate(engines):
            e.set_x(X, bounds[r])
            e.covariance()
            e.set_w(None)
        rows = ppr * T
        for _ in range(n_iter):
            for e in engines:
                e.power()
            for r in range(world):                       # the all-gather: rank-major concatenation
                gathered[r * rows:(r + 1) * rows].copy_(local[r], non_blocking=True)
            for e in engines:
                e.update(gathered)
        W = np.concatenate([e.plan.get_w(np.complex128) for e in engines], axis=0)
        Y = np.concatenate([e.demix(True) for e in engines], axis=1)
        for e in engines:
            e.close()
    return W, Y


@pytest.mark.parametrize("mode", ["precise", "fast"])
@pytest.mark.parametrize("model", ["laplace", "gauss"])
@pytest.mark.parametrize("bounds", [(0, 256, 512), (0, 128, 256, 384, 512), (0, 128, 256, 384, 512, 640, 768, 896, 1024), (0, 576, 1152)],
                         ids=["2-equal", "4-equal", "8-equal", "2-equal-9-parts-each"])
def test_equal_aligned_shards_are_bitwise_equal_to_the_single_plan(bounds, model, mode):
    import overiva_amd as oa

    X = orc.synth_mixture(T, bounds[-1], M, K, seed=21)
    W1, Y1 = _run_single(oa, X, model, mode, 6)
    W2, Y2 = _run_sharded(oa, X, model, mode, 6, list(bounds))
    assert np.all(np.isfinite(W1))
    assert np.array_equal(W1, W2)
    assert np.array_equal(Y1, Y2)


@pytest.mark.parametrize("model", ["laplace", "gauss"])
@pytest.mark.parametrize("bounds", [(0, 128, 448), (0, 192, 256, 448)], ids=["2-uneven", "3-uneven"])
def test_unequal_aligned_shards_agree_to_rounding(bounds, model):
    """unequal shards are padded with zero parts to one message size, which shifts the blocks of the canonical sum"""
    import overiva_amd as oa

    X = orc.synth_mixture(T, F, M, K, seed=21)
    W1, Y1 = _run_single(oa, X, model, "precise", 6)
    W2, Y2 = _run_sharded(oa, X, model, "precise", 6, list(bounds))
    assert orc.rel_err(W2, W1) < 1e-5 and orc.rel_err(Y2, Y1) < 1e-5


@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_unaligned_shards_agree_to_rounding(model):
    """boundaries inside a 64-bin batch (what shard_bounds gives for world sizes that do not divide the batches)"""
    import overiva_amd as oa
    from overiva_amd.sharded import shard_bounds

    X = orc.synth_mixture(T, F, M, K, seed=22)
    W1, Y1 = _run_single(oa, X, model, "precise", 6)
    W2, Y2 = _run_sharded(oa, X, model, "precise", 6, shard_bounds(F, 3))
    assert orc.rel_err(W2, W1) < 1e-5 and orc.rel_err(Y2, Y1) < 1e-5
    _, Wr = orc.overiva_staged(X, n_src=K, n_iter=6, proj_back=False, model=model, return_filters=True)
    assert orc.rel_err(W2, Wr) < 1e-5
