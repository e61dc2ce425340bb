"""The bin-sharded product path with REAL processes: N ranks, one HipEngine each, all on the one GPU of the test box,
collectives over gloo (RCCL does not accept two ranks on one device).  Everything but the transport of the all-gather is
what runs on an 8-GPU node: shard bounds, per-rank plans with F_total > F, the padded parts layout, the rank-order
activation sum, the gathers of Y / W, the callback cadence.  Compared with the single-process result."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import need  # noqa: F401  (keeps the helper importable the same way as the other GPU tests)

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, world, T, F, M, K, model, precision, n_iter, port, exchange="collective", init="eye"):
    out = str(tmp_path / f"sharded_{world}.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "helpers", "sharded_worker.py"), out, str(T), str(F), str(M),
           str(K), model, precision, str(n_iter), exchange, init]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("exchange", ["collective", "push"])
@pytest.mark.parametrize("world,F,model,precision", [(2, 128, "laplace", "precise"), (2, 128, "gauss", "fast"),
                                                     (3, 200, "laplace", "precise")])
def test_processes_sharing_one_gpu(tmp_path, world, F, model, precision, exchange):
    import overiva_amd as oa
    from oracle import overiva_oracle as orc

    T, M, K, n_iter = 300, 4, 2, 12
    got = _run(tmp_path, world, T, F, M, K, model, precision, n_iter, 29600 + world + (10 if exchange == "push" else 0), exchange)
    assert int(got["world"]) == world
    oa.set_precision(precision)
    os.environ["OIVA_RESIDENT"] = "0"        # the ranks run the four-launch path: compare with the same path in one process
    try:
        X = orc.synth_mixture(T, F, M, K, seed=11)
        seen = []
        Y, W = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True, model=model, return_filters=True,
                          callback=lambda y: seen.append(y.copy()))
    finally:
        oa.set_precision("auto")
        os.environ.pop("OIVA_RESIDENT", None)
    assert got["cb"].shape == np.stack(seen).shape
    if F % (64 * world) == 0:        # shard boundaries on 64-bin batches: the same bits as one process
        assert np.array_equal(got["W"], W) and np.array_equal(got["Y"], Y) and np.array_equal(got["cb"], np.stack(seen))
    else:                            # a batch straddles a boundary: the activation sums in another grouping
        assert orc.rel_err(got["W"], W) < 1e-5 and orc.rel_err(got["Y"], Y) < 1e-5


def test_sharded_init_eig_runs_on_every_ranks_device(tmp_path):
    """init_eig (overiva.py:106-109) with the bins sharded: every rank runs the device eigensolver on its own shard;
    same bits as one process (the eigenvectors are per bin, nothing is exchanged for them)"""
    import overiva_amd as oa
    from oracle import overiva_oracle as orc

    T, F, M, K, n_iter = 300, 128, 4, 2, 6
    got = _run(tmp_path, 2, T, F, M, K, "laplace", "mixed", n_iter, 29631, "collective", "eig")
    oa.set_precision("mixed")
    os.environ["OIVA_RESIDENT"] = "0"
    try:
        X = orc.synth_mixture(T, F, M, K, seed=11)
        Y, W = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True, return_filters=True, init_eig=True)
    finally:
        oa.set_precision("auto")
        os.environ.pop("OIVA_RESIDENT", None)
    assert np.array_equal(got["W"], W) and np.array_equal(got["Y"], Y)


@pytest.mark.parametrize("exchange", ["collective", "push"])
def test_bench_two_ranks_on_one_gpu(exchange):
    """bench.py's N > 1 path with two real ranks (both on GPU 0, gloo transport): one JSON line from rank 0, as the last
    line of stdout, with the contract fields, a per-rank stage breakdown for both ranks and a finite value"""
    import json

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611" if exchange == "collective" else "29612", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--backend", "gloo",
           "--single-device", "--exchange", exchange]
    r = subprocess.run(cmd, cwd=REPO, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert sum(1 for l in lines if l.startswith("{")) == 1
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"] is None
    ranks = d["ranks"]["per_rank_stage_ms"]
    assert d["ranks"]["exchange"] == exchange
    assert [x["rank"] for x in ranks] == [0, 1] and ranks[0]["bins"] == [0, 1024] and ranks[1]["bins"] == [1024, 2048]
