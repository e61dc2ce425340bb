"""The exchange of the ranks' partial source powers INSIDE the activation kernel of the four-launch iteration
(csrc/kernels_misc.hip::activation_xchg_kernel, overiva.py:152-155 with the bins sharded over GPUs whose shards do not
fit on chip: 2 and 4 GPUs at the headline shape): no collective, no host in the loop, graphs of four kernels replayed.
One GPU here: the loop-back form (this GPU plays every rank, the other ranks' sums are exact zeros -- the result must be
the single-GPU one BIT FOR BIT, eager and from graphs), a rank that does not deliver, and real processes sharing the GPU."""
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


def _run(oa, X, K, model, mode, chunks, world=0, graph=False):
    T, F, M = X.shape
    with oa.Plan(T, F, M, K, model) as p:
        p.set_precision(mode)
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        if world:
            p.fused_loopback(world)
        if graph:
            p.use_graph(True)
        for n in chunks:
            p.iterate(n)
        p.sync()
        return p.get_w(np.complex128), p.demix(False)


@pytest.mark.parametrize("shape", [(4000, 512, 8, 2), (1000, 513, 4, 2), (300, 70, 6, 3), (257, 64, 16, 16), (200, 40, 8, 1)])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_loopback_is_bit_equal_to_one_gpu(oa, shape, world):
    """this GPU as every rank: the rank's own parts are added in the order of the plain activation kernel and the other
    ranks contribute exact zeros, so W and Y are those of the plain four-launch path bit for bit -- eager launches and
    graph replay (the epoch is counted on the device), iterations in several calls, both models"""
    T, F, M, K = shape
    X = orc.synth_iid(T, F, M, seed=21)
    for model in ("laplace", "gauss"):
        W0, Y0 = _run(oa, X, K, model, "mixed", [3, 9, 1])
        for graph in (False, True):
            W, Y = _run(oa, X, K, model, "mixed", [3, 9, 1], world=world, graph=graph)
            assert np.array_equal(W, W0) and np.array_equal(Y, Y0), (model, graph)


def test_a_rank_that_does_not_deliver_is_reported(oa):
    T, F, M, K = 300, 64, 4, 2
    X = orc.synth_iid(T, F, M, seed=2)
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.fused_loopback(2)
        p.iterate(2)
        p.sync()
        p.fused_debug(timeout_ms=20, stall=True)
        p.iterate(1)
        with pytest.raises(RuntimeError, match="gave up waiting"):
            p.sync()


def test_switching_the_exchange_off_restores_the_plain_path(oa):
    T, F, M, K = 300, 64, 4, 2
    X = orc.synth_iid(T, F, M, seed=2)
    W0, _ = _run(oa, X, K, "laplace", "mixed", [4])
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision("mixed")
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.use_graph(True)
        p.fused_loopback(4)
        p.iterate(2)
        p.fused_loopback(0)
        p.iterate(2)
        assert np.array_equal(p.get_w(np.complex128), W0)


def _workers(tmp_path, world, T, F, M, K, model, precision, n_iter, port):
    out = str(tmp_path / f"fused_{world}.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "helpers", "sharded_worker.py"), out, str(T), str(F), str(M),
           str(K), model, precision, str(n_iter), "fused", "eye", "gloo", "mixture"]
    r = subprocess.run(cmd, cwd=REPO, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("world,F,model", [(2, 128, "laplace"), (3, 200, "gauss"), (4, 256, "laplace"), (8, 512, "laplace"),
                                           (2, 1024, "laplace"), (2, 384, "gauss"), (4, 2048, "laplace")])
def test_processes_sharing_one_gpu(oa, tmp_path, world, F, model):
    """real processes, one plan each on the box's one GPU, the exchange inside their activation kernels through IPC-mapped
    fine-grained buffers (across GPUs the stores travel over xGMI): callbacks, projection back and the gathers of the sharded
    driver around graph replays.  Against the four-launch path in ONE process: equal shards made of whole blocks of the
    canonical sum (1024 bins on 2 ranks: 4 blocks of 2 parts each; 2048 on 4: 2 blocks of 4) send their block sums, and the
    result is the single-GPU one BIT FOR BIT; unequal shards (200 bins on 3 ranks) send one sum per rank and agree to rounding."""
    T, M, K, n_iter = 300, 4, 2, 12
    got = _workers(tmp_path, world, T, F, M, K, model, "mixed", n_iter, 29700 + world)
    assert int(got["world"]) == world and str(got["exchange"]) == "fused", (str(got["exchange"]), str(got["refused"]))
    oa.set_precision("mixed")
    os.environ["OIVA_RESIDENT"] = "0"
    try:
        X = orc.synth_mixture(T, F, M, K, seed=11)
        seen = []
        Y, W = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True, model=model, return_filters=True, callback=lambda y: seen.append(y.copy()))
    finally:
        oa.set_precision("auto")
        os.environ.pop("OIVA_RESIDENT", None)
    from overiva_amd.sharded import fused_blocks, shard_bounds

    eW, eY = orc.rel_err(got["W"], W), orc.rel_err(got["Y"], Y)
    b = shard_bounds(F, world)
    whole_blocks = len({b[r + 1] - b[r] for r in range(world)}) == 1 and (F // world) % (64 * -(-(-(-F // 64)) // 8)) == 0
    print(f"\n[fused, {world} processes, {F} bins, {fused_blocks(b)} block(s) per rank] W {eW:.1e} Y {eY:.1e}")
    assert got["cb"].shape == np.stack(seen).shape
    if whole_blocks:
        assert np.array_equal(got["W"], W) and np.array_equal(got["Y"], Y) and np.array_equal(got["cb"], np.stack(seen))
    else:
        assert eW < 2e-5 and eY < 2e-5


def test_bench_under_an_external_launcher_without_the_ipc_variable():
    """`torch.distributed.run ... bench.py --gpus 2` started by SOMEBODY ELSE, with HSA_ENABLE_IPC_MODE_LEGACY
    absent from the parent's environment: bench.py (and `import overiva_amd`) put it there before their first GPU call, so
    the ranks can export their gather buffers and `--exchange auto` resolves to the in-kernel exchange (2 ranks x 1024 bins do
    not fit on chip: `fused`), not silently to the collective; the line says so at the top level"""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "OIVA_EXCHANGE")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29721", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--backend", "gloo",
           "--single-device"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ranks"]["exchange_requested"] == "auto"
    assert d["ranks"]["exchange"] == "fused" and d["exchange_degraded"] is None, (d["ranks"], r.stderr[-3000:])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["graph"] is True


def test_save_and_restore_of_the_demixing_state(oa):
    """oiva_plan_save_w / oiva_plan_restore_w: what a caller of the in-kernel exchanges falls back on -- after a wait that gave
    up the plan reports it from every reader (sync, get_w, demix), disconnecting clears the condition, the restored state
    continues on the plain path to the bits of a run that never used the exchange"""
    T, F, M, K = 300, 64, 4, 2
    X = orc.synth_iid(T, F, M, seed=2)
    W0, _ = _run(oa, X, K, "laplace", "mixed", [5])
    with oa.Plan(T, F, M, K, "laplace") as p:
        p.set_precision("mixed")
        p.set_x(X)
        p.covariance()
        p.set_w(None)
        p.fused_loopback(2)
        p.iterate(2)
        p.sync()
        p.save_w()
        p.fused_debug(timeout_ms=20, stall=True)
        p.iterate(3)
        for reader in (p.sync, p.get_w, lambda: p.demix(False)):
            with pytest.raises(RuntimeError, match="gave up waiting"):
                reader()
        p.fused_debug(timeout_ms=0, stall=False)
        p.fused_loopback(0)                 # off: the condition is history
        p.restore_w()
        p.iterate(3)
        p.sync()
        assert np.array_equal(p.get_w(np.complex128), W0)
