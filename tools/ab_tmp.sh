for lib in libprev_dev.so dev libprev_dev.so dev; do AB_REPS=4 AB_TAG="$lib" AM_HIP_LIBRARY=$lib timeout 300 python tools/wide_bench.py 2>&1 | tail -1 | cut -c1-120; done
for lib in libprev_dev.so dev; do AB_K=10 AB_REPS=3 AB_TAG="$lib" AM_HIP_LIBRARY=$lib timeout 300 python tools/wide_bench.py 2>&1 | tail -1 | cut -c1-230; done
