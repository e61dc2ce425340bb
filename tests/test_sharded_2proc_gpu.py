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


def _run(tmp_path, world, T, F, M, K, model, precision, n_iter, port, exchange="collective", init="eye", backend="gloo", data="mixture"):
    out = str(tmp_path / f"sharded_{world}.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "helpers", "sharded_worker.py"), out, str(T), str(F), str(M),
           str(K), model, precision, str(n_iter), exchange, init, backend, data]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
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


def test_auxiva_pca_with_sharded_bins(tmp_path):
    """auxiva_pca() while bin sharding is on: the PCA front end runs on every rank's GPU over all bins, the inner determined
    solve shards the reduced tensor (a host array: a device-resident tensor cannot be sharded); same result as one process"""
    import overiva_amd as oa
    from oracle import overiva_oracle as orc

    T, F, M, K, n_iter = 300, 128, 4, 2, 6
    got = _run(tmp_path, 2, T, F, M, K, "laplace", "mixed", n_iter, 29633, "collective", "pca")
    oa.set_precision("mixed")
    os.environ["OIVA_RESIDENT"] = "0"
    try:
        Y = oa.auxiva_pca(orc.synth_mixture(T, F, M, K, seed=11), n_src=K, n_iter=n_iter, proj_back=True, model="laplace")
    finally:
        oa.set_precision("auto")
        os.environ.pop("OIVA_RESIDENT", None)
    assert got["Y"].shape == Y.shape and orc.rel_err(got["Y"], Y) < 1e-5


def test_resident_shard_through_an_rccl_group(tmp_path):
    """one rank's shard of the headline shape at 8 GPUs (256 x 4000 x 8 / 2) through the bin-sharded driver over a REAL
    RCCL process group (backend nccl, world = 1: what this 1-GPU box can hold) with the X-resident kernel: callbacks,
    projection back and the gathers of the sharded path around one persistent launch per 10 iterations; against the
    single-process drop-in call"""
    import overiva_amd as oa
    from oracle import overiva_oracle as orc

    T, F, M, K, n_iter = 4000, 256, 8, 2, 12
    got = _run(tmp_path, 1, T, F, M, K, "laplace", "mixed", n_iter, 29641, "resident", "eye", "nccl", "iid")
    assert str(got["backend"]) == "nccl" and int(got["world"]) == 1 and bool(got["resident"]), str(got["refused"])
    X = orc.synth_iid(T, F, M, seed=11)
    oa.set_precision("mixed")
    try:
        seen = []
        Y, W = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True, return_filters=True, callback=lambda y: seen.append(y.copy()))
    finally:
        oa.set_precision("auto")
    assert oa.last_solver_info()["resident_launches"] >= 2
    assert np.array_equal(got["W"], W) and np.array_equal(got["Y"], Y) and np.array_equal(got["cb"], np.stack(seen))
    _, Wr = orc.overiva_staged(X.astype(np.complex128), n_src=K, n_iter=n_iter, proj_back=False, return_filters=True)
    assert orc.rel_err(W, Wr) < 1e-5


@pytest.mark.parametrize("world,F,T", [(8, 128, 300), (8, 512, 128)])
def test_eight_resident_ranks_on_one_gpu(tmp_path, world, F, T):
    """EIGHT real processes -- the world size of BASELINE configs[3] -- whose persistent kernels are resident side by side on
    the box's one GPU (16 and 64 bins per rank: 10 and 32 workgroups each) and exchange their column sums through all eight
    slots of every rank's IPC-mapped gather buffer, every workgroup adding the eight ranks' sums in rank order: the code path
    of `bench.py --gpus 8`, minus the flight over xGMI.  Same result as the four-launch path in one process to rounding (the
    sums cross the ranks as tagged words that give one mantissa bit to the epoch, so no grouping of the bins makes the two
    bit-identical; the 64-bin case has the shard bounds on the 64-bin batches of the single-GPU sum)."""
    import overiva_amd as oa
    from oracle import overiva_oracle as orc

    M, K, n_iter = 4, 2, 12
    got = _run(tmp_path, world, T, F, M, K, "laplace", "mixed", n_iter, 29660 + F // 128, "resident")
    assert int(got["world"]) == world and bool(got["resident"]), str(got["refused"])
    oa.set_precision("mixed")
    os.environ["OIVA_RESIDENT"] = "0"
    try:
        X = orc.synth_mixture(T, F, M, K, seed=11)
        seen = []
        Y, W = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True, return_filters=True, callback=lambda y: seen.append(y.copy()))
    finally:
        oa.set_precision("auto")
        os.environ.pop("OIVA_RESIDENT", None)
    eW, eY = orc.rel_err(got["W"], W), orc.rel_err(got["Y"], Y)
    print(f"\n[resident, {world} processes, {F} bins] W {eW:.1e} Y {eY:.1e}")
    assert eW < 2e-5 and eY < 2e-5 and got["cb"].shape == np.stack(seen).shape


@pytest.mark.parametrize("world,F", [(2, 128), (3, 192)])
def test_resident_exchange_between_processes_sharing_one_gpu(tmp_path, world, F):
    """the exchange INSIDE the X-resident kernel with real processes: every rank's persistent kernel stores its column sums
    into every rank's gather buffer (IPC-mapped fine-grained memory; across GPUs these stores travel over xGMI) and waits for
    the others' -- here the ranks' kernels share the one GPU of the box, small enough to be resident side by side.  Same
    result as the four-launch path in one process (to rounding: gamma is applied after the sums, the parts lose one
    mantissa bit to the epoch tag)."""
    import overiva_amd as oa
    from oracle import overiva_oracle as orc

    T, M, K, n_iter = 300, 4, 2, 12
    got = _run(tmp_path, world, T, F, M, K, "laplace", "mixed", n_iter, 29650 + world, "resident")
    assert int(got["world"]) == world and bool(got["resident"]), str(got["refused"])
    oa.set_precision("mixed")
    os.environ["OIVA_RESIDENT"] = "0"
    try:
        X = orc.synth_mixture(T, F, M, K, seed=11)
        seen = []
        Y, W = oa.overiva(X, n_src=K, n_iter=n_iter, proj_back=True, return_filters=True, callback=lambda y: seen.append(y.copy()))
    finally:
        oa.set_precision("auto")
        os.environ.pop("OIVA_RESIDENT", None)
    eW, eY = orc.rel_err(got["W"], W), orc.rel_err(got["Y"], Y)
    print(f"\n[resident, {world} processes] W {eW:.1e} Y {eY:.1e}")
    assert eW < 2e-5 and eY < 2e-5 and got["cb"].shape == np.stack(seen).shape


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


def test_bench_starts_its_own_ranks_and_runs_the_resident_exchange():
    """`python bench.py --gpus 2` WITHOUT a launcher (no WORLD_SIZE): the ranks are children started before any GPU call, the
    parent relays rank 0's line.  A shape small enough for the two ranks' persistent kernels to be resident side by side on
    the box's one GPU (one workgroup per CU each), the exchange inside the X-resident kernels"""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--config", "tiny", "--steps", "20", "--warmup", "3",
           "--backend", "gloo", "--single-device", "--exchange", "resident", "--launch-timeout", "150"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["launcher"]["spawned_ranks"] == 2 and d["launcher"]["attempts"][-1]["status"] == "ok"
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0
    assert d["ranks"]["exchange"] == "resident" and d["ranks"]["fallback"] is None, (d["ranks"], d["launcher"], r.stderr[-3000:])
    assert len(d["ranks"]["per_rank_stage_ms"]) == 2 and "resident_phase_us_workgroup0" in d["ranks"]["per_rank_stage_ms"][0]


@pytest.mark.parametrize("world,exchange,config,port", [(4, "fused", "headline", "29621"), (8, "resident", "tiny", "29622")])
def test_bench_four_and_eight_ranks_on_one_gpu(world, exchange, config, port, tmp_path):
    """bench.py's N > 1 path as the driver would launch it on a node (torch.distributed.run, one process per rank;
    here all on GPU 0 over gloo) with FOUR ranks through the exchange inside the activation kernel (the headline shape: 512 bins
    per rank) and EIGHT through the exchange inside the X-resident kernel: not degraded, every rank in the group, and the
    demixing matrices after three iterations are those of the single-GPU run.  (To rounding, not to the bit, at this shape: the
    activations r are the same bits at any number of equal shards, but a 512-bin plan splits the 4000 frames of the covariance
    pass into more float32 chains than the 2048-bin plan does; where the plans split alike the bits agree: tests/test_sharded_gpu.py.)"""
    import json

    base = ["--config", config, "--steps", "6", "--warmup", "2", "--w-digest", "3", "--w-dump", str(tmp_path)]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(REPO, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--single-device", "--exchange", exchange] + base
    r = subprocess.run(cmd, cwd=REPO, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] > 0 and d["exchange_degraded"] is None, (d.get("exchange_degraded"), d["ranks"])
    assert d["ranks"]["rccl_ranks"] == world and d["ranks"]["exchange"] == exchange and d["ranks"]["fallback"] is None
    per_rank = d["ranks"]["per_rank_stage_ms"]
    assert [x["rank"] for x in per_rank] == list(range(world)) and all(len(x["w_digest"]) == 16 for x in per_rank)
    assert d["untimed_pre_pass_steps"] >= 6
    one = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--no-cpu", "--no-configs", "--no-other-mode", "--digest-shards", str(world)] + base,
                         cwd=REPO, capture_output=True, text=True, timeout=500)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    s = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert len(s["w_digest_shards"]["sha256_16"]) == world
    from oracle import overiva_oracle as orc

    W1 = np.load(tmp_path / "w_rank0of1.npy")
    Wn = np.concatenate([np.load(tmp_path / f"w_rank{k}of{world}.npy") for k in range(world)], axis=0)
    e = orc.rel_err(Wn, W1)
    print(f"\n[bench --gpus {world}, {exchange}] W after 3 iterations vs the single-GPU run: {e:.1e}")
    assert Wn.shape == W1.shape and e < 2e-6
