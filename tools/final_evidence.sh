#!/bin/bash
# The round's whole GPU evidence from ONE library build, in the order that lets every record agree with the others
# (run through gpurun from the repo root):  tools/final_evidence.sh r5
#   1. tools/profile_bench.sh (trace + three --pmc passes, default workload and --dim 128) and the traffic table derived
#      from them, written where bench.py reads it (profiles/traffic.json ON THE BOX: the bench line of step 2 then carries
#      `roofline.traffic` of this very build instead of STALE) - tools/collect_profiles.sh rebuilds the same table at home;
#   2. tools/run_round_checks.sh (GPU suite, bench.py, multi-rank launch at full size, configs[4]);
#   3. the scale model, the fuzzers (FUZZ=0 skips them), the f64 and the one-call-per-rank probes.
R=${1:-r5}
mkdir -p gpurun_out/$R
bash tools/profile_bench.sh $R > gpurun_out/$R/profile_bench.log 2>&1
bash tools/profile_bench.sh ${R}_d128 --dim 128 > gpurun_out/$R/profile_bench_d128.log 2>&1
rm -f profiles/traffic.json
python3 tools/update_traffic.py gpurun_out/prof_$R --table profiles/traffic.json > /dev/null 2>&1
sed -i "s#gpurun_out/prof_$R#profiles/$R#" profiles/traffic.json
bash tools/run_round_checks.sh $R 2>&1 | tail -12
timeout 600 python tools/scale_model.py > gpurun_out/$R/scale_model.json 2> gpurun_out/$R/scale_model.err
timeout 300 python tools/f64_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/$R/f64_probe.txt
timeout 300 python tools/sharded_c_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/$R/sharded_c_probe.txt
if [ "${FUZZ:-1}" != "0" ]; then
  timeout 1200 python tools/fuzz_filter.py 20 111 2>&1 | grep -v amdgpu.ids > gpurun_out/$R/fuzz_filter_20_cases.txt
  FUZZ_ROWS=150000,200000 FUZZ_DIMS=128,512 timeout 600 python tools/fuzz_filter.py 2 113 2>&1 | grep -v amdgpu.ids > gpurun_out/$R/fuzz_filter_large.txt
  timeout 600 python tools/fuzz_part.py 10 112 2>&1 | grep -v amdgpu.ids > gpurun_out/$R/fuzz_part_10_cases.txt
  tail -qn 1 gpurun_out/$R/fuzz_*.txt
fi
head -4 gpurun_out/prof_$R/trace.summary.txt; cat gpurun_out/$R/sharded_c_probe.txt
