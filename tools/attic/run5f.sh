mkdir -p gpurun_out/r5f
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_demix_io_gpu.py tests/test_oneshot_gpu.py -m gpu -q -x -k "overiva_matches or demix_io or slabs or hand_over or headline_size_against or complex128_in or not_mutated or auxiva or oneshot or two_plans" 2>&1 | tail -6 > gpurun_out/r5f/pytest.log
timeout 200 python tools/e2e_host.py > gpurun_out/r5f/e2e_cache.log 2>&1
OIVA_PLAN_CACHE=0 timeout 200 python tools/e2e_host.py > gpurun_out/r5f/e2e_nocache.log 2>&1
tail -3 gpurun_out/r5f/pytest.log
