mkdir -p gpurun_out/r5j
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg5 or ill_conditioned or odd_shapes or random_shapes or ip_update or overiva_matches" 2>&1 | tail -6 > gpurun_out/r5j/pytest.log
timeout 300 python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu --no-configs --no-other-mode > gpurun_out/r5j/cfg5_solve.json 2> gpurun_out/r5j/cfg5_solve.err
OIVA_DET16_INVERSE=1 timeout 300 python bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu --no-configs --no-other-mode > gpurun_out/r5j/cfg5_inverse.json 2> gpurun_out/r5j/cfg5_inverse.err
tail -3 gpurun_out/r5j/pytest.log
