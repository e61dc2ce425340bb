# round 5: everything profiles/r05_* is made from (about 20 minutes of box time)
mkdir -p gpurun_out/prof
rm -f gpurun_out/parity.jsonl
OIVA_PARITY_LOG=$PWD/gpurun_out/parity.jsonl timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/prof/pytest_full.log
timeout 1800 bash tools/collect_profiles.sh > gpurun_out/prof/collect.log 2>&1
tail -3 gpurun_out/prof/pytest_full.log; tail -2 gpurun_out/prof/collect.log
