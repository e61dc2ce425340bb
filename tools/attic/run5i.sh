mkdir -p gpurun_out/r5i
timeout 900 python -m pytest tests/test_resident_gpu.py tests/test_sharded_2proc_gpu.py tests/test_sharded_gpu.py tests/test_fused_exchange_gpu.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r5i/pytest.log
timeout 300 python tools/shard_step.py > gpurun_out/r5i/shard_step.log 2>&1
tail -3 gpurun_out/r5i/pytest.log; cat gpurun_out/r5i/shard_step.log
