mkdir -p gpurun_out/r5e
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_demix_io_gpu.py -m gpu -q -x -k "power_pass or demix_power or cfg5 or demix_io or slabs or hand_over or ill_conditioned or ip_update" 2>&1 | tail -6 > gpurun_out/r5e/pytest.log
timeout 200 python tools/e2e_host.py > gpurun_out/r5e/e2e_a.log 2>&1
timeout 200 python tools/e2e_host.py > gpurun_out/r5e/e2e_b.log 2>&1
timeout 300 python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu --no-configs --no-other-mode > gpurun_out/r5e/cfg5_lds.json 2> gpurun_out/r5e/cfg5_lds.err
OIVA_POWER_LDS=0 timeout 300 python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu --no-configs --no-other-mode > gpurun_out/r5e/cfg5_old.json 2> gpurun_out/r5e/cfg5_old.err
tail -3 gpurun_out/r5e/pytest.log
