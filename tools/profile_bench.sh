#!/bin/bash
# The rocprofv3 evidence for bench.py from ONE library build (run through gpurun from the repo root):
#   kernel trace + stats, three PMC passes (separate runs, counters only: MFMA busy / LDS, FETCH_SIZE, WRITE_SIZE + L2 hits),
#   their per-kernel summaries, and the traffic table derived from them - every file of the directory carries the same
#   library stamp (sha256 over the sources the loaded .so was built from).
# Usage: tools/profile_bench.sh <tag> [extra bench.py arguments, e.g. --dim 128]     -> gpurun_out/prof_<tag>/...
# Then copy the directory's *.summary.txt, library.stamp.json and traffic.json to profiles/<round>/ (and traffic.json to
# profiles/traffic.json, the one bench.py reads); tools/update_traffic.py refuses a table that mixes stamps.
set -u
TAG=${1:-r5}
shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
STAMP=audio-metrics_amd/lib/libaudio_metrics_hip.so.stamp.json
cp "$STAMP" "$OUT/library.stamp.json"
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-process-cold $*"
echo "$ARGS" > "$OUT/command.txt"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python3 $ARGS > "$OUT/trace_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
    -d "$OUT/pmc_sq" -o bench -- python3 $ARGS > "$OUT/pmc_sq_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d "$OUT/pmc_fetch" -o bench -- python3 $ARGS > "$OUT/pmc_fetch_stdout.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_write" -o bench -- python3 $ARGS > "$OUT/pmc_write_stdout.log" 2>&1
for pass in trace pmc_sq pmc_fetch pmc_write; do
    db=$(find "$OUT/$pass" -name "*.db" | head -1)
    if [ -n "$db" ]; then
        { echo "# library sources_sha256 $(python3 -c "import json; print(json.load(open('$OUT/library.stamp.json'))['sources_sha256'])")   command: python3 $ARGS"
          python3 tools/rocpd_summary.py "$db"; } > "$OUT/$pass.summary.txt" 2>&1
    fi
done
# one build from the first pass to the last (a rebuild in between would mix two libraries in one directory)
if ! cmp -s "$STAMP" "$OUT/library.stamp.json"; then
    echo "the library changed while profiling: $OUT is not a one-build profile" >&2
    exit 1
fi
python3 tools/update_traffic.py "$OUT" --table "$OUT/traffic.json"
# the raw rocprofv3 databases (tens of MB per pass) stay on the box: gpurun copies back at most 64 MiB in all
rm -rf "$OUT/trace" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write"
ls -la "$OUT"
