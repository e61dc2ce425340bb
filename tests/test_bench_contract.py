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
