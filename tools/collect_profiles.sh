#!/bin/bash
# GPU box: everything profiles/rNN_* is made from.  Usage: bash tools/collect_profiles.sh   (writes gpurun_out/prof/)
# Counters are collected in their own passes with --kernel-trace only (no --stats, no API traces), 4 counters per pass.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
HEAD="--steps 6 --warmup 2 --no-cpu --no-other-mode --no-configs --graph 0"

# bench lines (the default run carries the CPU baseline)
$B > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
$B --config cfg5 --no-cpu > "$OUT/bench_cfg5.json" 2> "$OUT/bench_cfg5.err"
$B --force-sharded --no-cpu --no-other-mode > "$OUT/bench_sharded_1rank.json" 2> "$OUT/bench_sharded.err"

# kernel-trace statistics of the bench command
# (the headline workload alone, so that a kernel's average is the average over ITS launches of that workload: the secondary
#  configs run the same kernel instantiations on other shapes -- those go to stats_configs)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_headline" -o s -- $B --steps 20 --warmup 3 --no-cpu --no-configs > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_configs" -o s -- $B --steps 20 --warmup 3 --no-cpu --no-other-mode > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_cfg5" -o s -- $B --config cfg5 --steps 20 --warmup 3 --no-cpu > /dev/null 2>&1

# counters, headline shape, fast mode, eager launches
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_headline/p$i" -o p -- $B $HEAD > /dev/null 2>&1
done
# HBM traffic of the precise mode (float64 covariance on the vector ALU, float64 per-bin algebra)
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_precise/p$i" -o p -- $B $HEAD --precision precise > /dev/null 2>&1
done
# counters, cfg5 (default arithmetic: the covariance kernel with the sources on the fp32 matrix cores)
i=0
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_cfg5/p$i" -o p -- $B --config cfg5 $HEAD > /dev/null 2>&1
done
# 16 channels / 2 sources (the four-lanes-per-(bin, frame) covariance kernel), default arithmetic of that shape
$B --config m16k2 --no-cpu > "$OUT/bench_m16k2.json" 2> "$OUT/bench_m16k2.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_m16k2" -o s -- $B --config m16k2 --steps 20 --warmup 3 --no-cpu > /dev/null 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_m16k2/p$i" -o p -- $B --config m16k2 $HEAD > /dev/null 2>&1
done
# the X-resident kernel: HBM traffic per launch of 50 iterations (X once + the exchange words), shard of the headline shape and configs[1]
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_resident_shard8/p$i" -o p -- python3 $ROOT/tools/resident_case.py 4000 256 8 2 50 > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_resident_cfg2/p$i" -o p -- python3 $ROOT/tools/resident_case.py 1000 513 4 2 50 > /dev/null 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_resident" -o s -- python3 $ROOT/tools/resident_case.py 4000 256 8 2 50 > "$OUT/resident_shard8.log" 2>&1
# (round 4) the shard step at several launch lengths, single rank and loop-back world 8; the fixed cost of a launch; every
# workgroup's phase boundaries; the 2- and 4-GPU shards through the exchange inside the activation kernel (loop-back)
python3 $ROOT/tools/shard_step.py > "$OUT/shard_step.log" 2>&1
python3 $ROOT/tools/launch_cost.py > "$OUT/launch_cost.log" 2>&1
python3 $ROOT/tools/exp_resident_trace.py 4000 256 8 2 20 mixed > "$OUT/resident_trace_shard8.log" 2>&1
python3 $ROOT/tools/exp_resident_trace.py 4000 256 8 2 20 mixed 8 > "$OUT/resident_trace_shard8_loopback8.log" 2>&1
python3 $ROOT/tools/res_ab.py > "$OUT/resident_configs.log" 2>&1
python3 $ROOT/tools/cfg5_cov_sweep.py > "$OUT/cfg5_cov_sweep.log" 2>&1
# (round 5) the drop-in call with host arrays, end to end and phase by phase; the reference's sweep shapes stage by stage
python3 $ROOT/tools/e2e_phases.py > "$OUT/e2e_phases.log" 2>&1
python3 $ROOT/tools/e2e_host.py > "$OUT/e2e_host.log" 2>&1
python3 $ROOT/tools/stage_times_reference_shapes.py > "$OUT/stage_times_reference_shapes.log" 2>&1
# (round 5) the 2- / 4-GPU shards without and with the in-kernel exchange, alternating; the sweep shapes against forced frame splits;
# the achieved parity errors of every end-to-end row
python3 $ROOT/tools/shard_fused_ab.py > "$OUT/shard_fused_ab.log" 2>&1
python3 $ROOT/tools/sweep_shape_splits.py > "$OUT/sweep_shape_splits.log" 2>&1
(cd $ROOT && rm -f "$OUT/parity.jsonl" && OIVA_PARITY_LOG="$OUT/parity.jsonl" python3 -m pytest tests/test_gpu_parity.py -m gpu -q > "$OUT/parity_pytest.log" 2>&1)
# (round 6) configs[4]: the per-bin update in its two forms, the new kernel's roles alone and workgroup 0's timeline (variant builds of
# tools/build_variant.py, when present)
OIVA_DET16_ROWS=0 timeout 200 python3 $ROOT/tools/r6/det16_ab.py > "$OUT/det16_ab.log" 2>&1
timeout 200 python3 $ROOT/tools/r6/det16_ab.py >> "$OUT/det16_ab.log" 2>&1
for v in r16onlya r16onlyb; do
    [ -f $ROOT/overiva_amd/liboveriva_hip_$v.so ] && OIVA_LIB=$ROOT/overiva_amd/liboveriva_hip_$v.so timeout 200 python3 $ROOT/tools/r6/det16_roles.py >> "$OUT/det16_ab.log" 2>&1
done
[ -f $ROOT/overiva_amd/liboveriva_hip_r16trace.so ] && OIVA_LIB=$ROOT/overiva_amd/liboveriva_hip_r16trace.so timeout 200 python3 $ROOT/tools/r6/det16_trace.py > "$OUT/det16_trace.log" 2>&1
# keep what travels back small: drop the raw kernel traces of the --stats runs
find "$OUT" -name "*_kernel_trace.csv" -path "*stats_*" -delete
find "$OUT" -name "*_agent_info.csv" -delete
du -sh "$OUT"
