mkdir -p gpurun_out/r5h
timeout 280 python tools/e2e_phases.py > gpurun_out/r5h/phases.log 2>&1
tail -12 gpurun_out/r5h/phases.log
