cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/p2; mkdir -p $O
cd $R
timeout 300 python3 bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -o s -- python3 bench.py --config cfg5 --steps 20 --warmup 3 --no-cpu > /dev/null 2>&1
find $O/stats_cfg5 -name "*kernel_trace.csv" -delete
cat $O/bench_cfg5.json | head -c 1500
find $O/stats_cfg5 -name "*kernel_stats.csv" | head -1 | xargs head -12
