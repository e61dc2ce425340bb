mkdir -p gpurun_out/r5l
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_demix_io_gpu.py -m gpu -q -x -k "one_launch or plan_kept or graph_cache or headline" 2>&1 | tail -8 > gpurun_out/r5l/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-other-mode > gpurun_out/r5l/bench_fused.json 2> gpurun_out/r5l/bench_fused.err
OIVA_COV_UPDATE=0 python bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-other-mode > gpurun_out/r5l/bench_unfused.json 2> gpurun_out/r5l/bench_unfused.err
python bench.py --steps 64 --warmup 5 --no-cpu --no-configs --no-other-mode > gpurun_out/r5l/bench_fused64.json 2> /dev/null
OIVA_COV_UPDATE=0 python bench.py --steps 64 --warmup 5 --no-cpu --no-configs --no-other-mode > gpurun_out/r5l/bench_unfused64.json 2> /dev/null
tail -4 gpurun_out/r5l/pytest.log
