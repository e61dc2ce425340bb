mkdir -p gpurun_out/r5g
timeout 250 python tools/e2e_phases.py > gpurun_out/r5g/phases_cache.log 2>&1
OIVA_PLAN_CACHE=0 timeout 250 python tools/e2e_phases.py > gpurun_out/r5g/phases_nocache.log 2>&1
OIVA_POOL_MB=0 OIVA_PLAN_CACHE=0 timeout 250 python tools/e2e_phases.py > gpurun_out/r5g/phases_nopool.log 2>&1
tail -4 gpurun_out/r5g/phases_cache.log
