"""N > 1 control flow of the bin-sharded driver on CPU: world_size 2 and 3 over gloo.

The worker (tests/gloo_worker.py) plugs a test-only oracle-backed engine into
overiva_amd.sharded.BinShardedSolver, so what is exercised is the product's partitioning, the
all-gather of partial powers, the rank-order sum and the result gathering -- not any CPU arithmetic
path of the product (there is none)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from oracle import overiva_oracle as orc


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("model", ["laplace", "gauss"])
def test_sharded_equals_unsharded(tmp_path, world, model):
    n_iter = 4
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "gloo_worker.py"),
                                       str(tmp_path), model, str(n_iter)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()[-3000:]

    T, F, M, K = 60, 11, 4, 2
    X = orc.synth_iid(T, F, M, seed=11)
    rng = np.random.default_rng(12)
    W0 = np.eye(M, K)[None] + 0.1 * (rng.standard_normal((F, M, K)) + 1j * rng.standard_normal((F, M, K)))
    Yr, Wr = orc.overiva_staged(X.astype(np.complex128), n_src=K, n_iter=n_iter, proj_back=True, W0=W0,
                                model=model, return_filters=True)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    covered = []
    for o in outs:
        # every rank holds the full, identical result
        assert np.array_equal(o["Y"], outs[0]["Y"]) and np.array_equal(o["W"], outs[0]["W"])
        assert o["Y"].shape == (T, F, K) and o["W"].shape == (F, M, K) and o["Cx"].shape == (F, M, M)
        covered.append((int(o["f0"]), int(o["f1"])))
    assert sorted(covered) == [(F * r // world, F * (r + 1) // world) for r in range(world)]
    # float32 exchange of the partial powers is the only rounding difference
    assert orc.rel_err(outs[0]["W"], Wr) < 1e-5
    assert orc.rel_err(outs[0]["Y"], Yr) < 1e-5
    assert orc.rel_err(outs[0]["Cx"], orc.input_covariance(X.astype(np.complex128))) < 1e-6


@pytest.mark.parametrize("wdtype", ["c64", "c128"])
def test_singular_bin_on_one_rank_raises_on_every_rank(tmp_path, wdtype):
    """BinShardedSolver.get_w gathers before it judges: when ONE rank's engine reports a singular solve, every rank raises
    LinAlgError instead of waiting in the all-gather -- also when W travels as complex128 (the 'mixed' / 'precise' modes),
    where the failing rank's NaN placeholder must have the same element size as what the healthy ranks send"""
    world, port = 3, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", OIVA_TEST_SINGULAR_RANK="1", OIVA_TEST_WDTYPE=wdtype)
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "gloo_worker.py"), str(tmp_path), "laplace", "2"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()[-3000:]
    for rank in range(world):
        assert (tmp_path / f"rank{rank}.txt").read_text().startswith("LinAlgError"), rank


def test_a_fused_exchange_that_gives_up_falls_back_to_the_collective(tmp_path):
    """when the in-kernel exchange of ONE rank gives up, every rank restores the demixing matrices it saved in front
    of the call, leaves that exchange and repeats the iterations through the collective: same result as a run that never
    used it (the engine here is the oracle-backed stand-in; the product engine's save / restore is covered on the GPU)"""
    world, n_iter, port = 3, 4, _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", OIVA_TEST_FUSED_GIVES_UP="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "gloo_worker.py"), str(tmp_path), "laplace", str(n_iter)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()[-3000:]
    T, F, M, K = 60, 11, 4, 2
    X = orc.synth_iid(T, F, M, seed=11)
    rng = np.random.default_rng(12)
    W0 = np.eye(M, K)[None] + 0.1 * (rng.standard_normal((F, M, K)) + 1j * rng.standard_normal((F, M, K)))
    _, Wr = orc.overiva_staged(X.astype(np.complex128), n_src=K, n_iter=n_iter, proj_back=True, W0=W0, return_filters=True)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert np.array_equal(o["W"], outs[0]["W"]) and orc.rel_err(o["W"], Wr) < 1e-5
