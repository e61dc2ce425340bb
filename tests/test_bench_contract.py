"""bench.py's JSON line keeps the driver's contract (checked on CPU: only the line builder and the accounting
helpers run, nothing touches a GPU)."""
import json
import types

import bench


def test_result_line_fields():
    args = types.SimpleNamespace(steps=50, warmup=5, graph=1, precision="fast")
    line = bench.result_line(args, 1, 0.0125)
    json.dumps(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in line
    assert line["value"] == 50 / 0.0125 and abs(line["ms_per_step"] - 0.25) < 1e-12
    assert line["higher_is_better"] is True and line["scaling"] == "strong" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" in line["config"] and line["n_gpus"] == 1
    assert "2048" in line["metric"] and "4000" in line["metric"] and "8 mics" in line["metric"]
    assert line["config"]["precision"].startswith("fast")


def test_cfg5_is_a_config_not_the_default():
    assert bench.CONFIGS["headline"]["M"] == 8 and bench.CONFIGS["cfg5"]["M"] == 16 and bench.CONFIGS["cfg5"]["K"] == 16
    assert (bench.T, bench.F, bench.M, bench.K) == (4000, 2048, 8, 2)


def test_algorithmic_bytes_match_survey():
    # 8TFM + 4TK + 8FKM^2 at the headline shape
    assert bench.cov_algorithmic_bytes(4000, 2048, 8, 2) == 526417152
    assert bench.cov_algorithmic_bytes(1000, 513, 4, 2) == 16555328


def test_committed_traffic_profile_is_readable():
    traffic, src = bench.measured_traffic("cov_dma_kernel<8, 2>")
    assert traffic is not None and src.startswith("profiles/")
    assert 0.99 < traffic / bench.cov_algorithmic_bytes(4000, 2048, 8, 2) < 1.05


# ---- `python bench.py --gpus N` without a launcher: the ranks are children, the run cannot hang ------------------
def _fake_worker(tmp_path, body):
    path = tmp_path / "fake_bench.py"
    path.write_text("import json, os, sys, time\nrank = int(os.environ['RANK'])\nargv = sys.argv[1:]\n" + body)
    return str(path)


def _launch_args(**kw):
    d = dict(gpus=2, exchange="collective", launch_timeout=20, launch_grace=2)      # (bench.py defaults to --exchange auto)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_launcher_relays_rank0_line(tmp_path, capsys):
    script = _fake_worker(tmp_path, "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                                    "assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'\n"
                                    "print('noise on stdout')\n"
                                    "if rank == 0: print(json.dumps({'value': 1.5, 'n_gpus': 2, 'argv': argv}))\n")
    rc = bench.launch_ranks(_launch_args(), ["--gpus", "2", "--steps", "3"], script=script)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert rc == 0 and len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 1.5 and d["launcher"]["spawned_ranks"] == 2 and d["launcher"]["attempts"][0]["status"] == "ok"
    assert d["argv"] == ["--gpus", "2", "--steps", "3", "--exchange", "collective"]


def test_launcher_kills_a_hung_run_and_reports_failure(tmp_path, capsys):
    import time

    script = _fake_worker(tmp_path, "time.sleep(3600)\n")
    t0 = time.monotonic()
    rc = bench.launch_ranks(_launch_args(launch_timeout=3, launch_grace=1), ["--gpus", "2"], script=script)
    assert rc == 1 and time.monotonic() - t0 < 60
    assert not [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]


def test_launcher_retries_a_failed_push_attempt_with_the_collective(tmp_path, capsys):
    script = _fake_worker(tmp_path, "ex = argv[argv.index('--exchange') + 1]\n"
                                    "if ex == 'push': sys.exit(17)\n"
                                    "if rank == 0: print(json.dumps({'value': 2.0, 'exchange': ex}))\n")
    rc = bench.launch_ranks(_launch_args(exchange="push"), ["--gpus", "2", "--exchange", "push"], script=script)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    d = json.loads(lines[-1])
    assert rc == 0 and d["exchange"] == "collective"
    assert [a["exchange"] for a in d["launcher"]["attempts"]] == ["push", "collective"] and d["launcher"]["attempts"][0]["returncode"] != 0
