mkdir -p gpurun_out/r5m
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_launch or headline_size_prop or headline_size_against" 2>&1 | tail -8 > gpurun_out/r5m/pytest.log
python bench.py --steps 64 --warmup 5 --no-cpu --no-configs --no-other-mode > gpurun_out/r5m/bench_fused64.json 2> gpurun_out/r5m/bench_fused.err
OIVA_COV_UPDATE=0 python bench.py --steps 64 --warmup 5 --no-cpu --no-configs --no-other-mode > gpurun_out/r5m/bench_unfused64.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5m/pmc_$c -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu --no-other-mode --no-configs --graph 0 > /dev/null 2>&1; done
cd $GRAFT_REPO_ROOT; tail -4 gpurun_out/r5m/pytest.log
