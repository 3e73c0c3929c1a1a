#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   kernel trace + stats, then PMC passes (separate runs, counters only) for MFMA busy / LDS / HBM bytes.
# Usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r1}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python3 $ARGS > "$OUT/trace_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
    -d "$OUT/pmc_sq" -o bench -- python3 $ARGS > "$OUT/pmc_sq_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d "$OUT/pmc_fetch" -o bench -- python3 $ARGS > "$OUT/pmc_fetch_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_write" -o bench -- python3 $ARGS > "$OUT/pmc_write_stdout.log" 2>&1
for pass in trace pmc_sq pmc_fetch pmc_write; do
    db=$(find "$OUT/$pass" -name "*.db" | head -1)
    [ -n "$db" ] && python3 tools/rocpd_summary.py "$db" > "$OUT/$pass.summary.txt" 2>&1
done
cp audio-metrics_amd/lib/libaudio_metrics_hip.so.stamp.json "$OUT/library.stamp.json"   # what tools/update_traffic.py ties the counters to
ls -la "$OUT"
