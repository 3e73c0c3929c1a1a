#!/bin/bash
# After tools/profile_bench.sh / run_round_checks.sh ran on the GPU box (results merged into gpurun_out/): copies the summaries
# of ONE build into profiles/<round>/ (+ /d128), rebuilds the traffic tables from them (one stamp per table) and refreshes
# profiles/traffic.json and profiles/scale_model.json.   Usage: tools/collect_profiles.sh r5
set -eu
R=${1:-r5}
mkdir -p profiles/$R/d128
cp gpurun_out/prof_$R/{trace,pmc_sq,pmc_fetch,pmc_write}.summary.txt gpurun_out/prof_$R/library.stamp.json gpurun_out/prof_$R/command.txt profiles/$R/
cp gpurun_out/prof_${R}_d128/{trace,pmc_sq,pmc_fetch,pmc_write}.summary.txt gpurun_out/prof_${R}_d128/library.stamp.json gpurun_out/prof_${R}_d128/command.txt profiles/$R/d128/
sed -i "s#gpurun_out/prof_${R}_d128/#profiles/$R/d128/#; s#gpurun_out/prof_$R/#profiles/$R/#" profiles/$R/*.summary.txt profiles/$R/d128/*.summary.txt
rm -f profiles/traffic.json profiles/$R/traffic.json profiles/$R/d128/traffic.json
python3 tools/update_traffic.py profiles/$R --table profiles/$R/traffic.json | grep -E "pstat|wide_kernel" || true
python3 tools/update_traffic.py profiles/$R/d128 --table profiles/$R/d128/traffic.json | grep -E "pstat|wide_kernel" || true
cp profiles/$R/traffic.json profiles/traffic.json
if [ -s gpurun_out/$R/scale_model.json ]; then cp gpurun_out/$R/scale_model.json profiles/scale_model.json; cp gpurun_out/$R/scale_model.json profiles/$R/scale_model.json; fi
for f in bench_n1.json pytest_gpu.txt multi_rank_full_size_one_gpu.txt bench_e2e_10k_n1.json; do
  [ -s gpurun_out/checks_$R/$f ] && cp gpurun_out/checks_$R/$f profiles/$R/$f
done
python3 - <<PY
import json
s = json.load(open("profiles/$R/library.stamp.json"))["sources_sha256"]
b = json.load(open("profiles/$R/bench_n1.json"))
print("profiles stamp", s[:12], "| bench line library", (b.get("library_sources_sha256") or "?")[:12], "|", round(b["ms_per_step"], 2), "ms per step")
PY
