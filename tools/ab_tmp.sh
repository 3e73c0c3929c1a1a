for pre in 2 1 2 1; do AB_REPS=5 AM_FAST_PRE_ANY=$pre AM_HIP_LIBRARY=dev timeout 300 python tools/ab_cross.py 2>&1 | tail -1 | sed "s/^/pre $pre: /" | cut -c1-200; done
for pre in 2 1; do AB_REPS=3 AB_TAG="pre$pre" AM_FAST_PRE_ANY=$pre AM_HIP_LIBRARY=dev timeout 300 python tools/wide_bench.py 2>&1 | tail -1 | cut -c1-230; done
for pre in 2 1; do AB_DATA=clap AB_K=10 AB_REPS=3 AB_TAG="pre$pre" AM_FAST_PRE_ANY=$pre AM_HIP_LIBRARY=dev timeout 300 python tools/wide_bench.py 2>&1 | tail -1 | cut -c1-230; done
