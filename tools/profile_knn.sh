#!/bin/bash
# rocprofv3 trace + TCC/SQ counters for the k-NN radii path alone (tools/ab_knn.py).  Usage: tools/profile_knn.sh <tag>
# NOTE: one TCC pass holds at most 4 counter slots (FETCH_SIZE = 3, WRITE_SIZE = 2): oversubscribing aborts
# rocprofv3 and its finaliser then hangs - every pass runs under `timeout`.
set -u
TAG=${1:-knn}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp AB_REPS=2
run() { name=$1; shift; timeout 120 rocprofv3 "$@" -d "$OUT/$name" -o knn -- python3 tools/ab_knn.py > "$OUT/${name}_stdout.log" 2>&1;
        python3 tools/rocpd_summary.py "$OUT/$name/knn_results.db" > "$OUT/$name.summary.txt" 2>&1; rm -f "$OUT/$name/knn_results.db"; }
run trace --kernel-trace --stats
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum
run pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
