for bits in 0 2 3 4 5; do AB_REPS=3 AB_TAG="drop$bits" AM_HALF_DROP_BITS=$bits AM_HIP_LIBRARY=dev timeout 300 python tools/wide_bench.py 2>&1 | tail -1 | cut -c1-220; done
for bits in 0 3 4; do AB_DATA=clap AB_K=10 AB_REPS=2 AB_TAG="drop$bits" AM_HALF_DROP_BITS=$bits AM_HIP_LIBRARY=dev timeout 300 python tools/wide_bench.py 2>&1 | tail -1 | cut -c1-220; done
