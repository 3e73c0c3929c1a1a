#!/bin/bash
# Samples rocm-smi power / clock while the wide-kernel bench loops (is the chip at its power limit during the f16 filters?)
( for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|Socket" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_samples.txt &
SAMPLER=$!
AB_REPS=40 timeout 120 python tools/wide_bench.py 2>&1 | tail -1 | cut -c1-120
wait $SAMPLER
cat gpurun_out/power_samples.txt | cut -c1-260 | head -14
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
