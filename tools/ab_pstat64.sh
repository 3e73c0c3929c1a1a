export AM_HIP_LIBRARY=dev AB_REPS=5
for d in 128 64; do
  for on in 0 1 0 1; do
    AM_PSTAT64=$on AB_DIM=$d AB_TAG=d$d-p64=$on timeout 300 python tools/wide_bench.py 2>&1 | tail -1
  done
done
AM_PSTAT64=0 AB_DIM=128 AB_K=10 AB_TAG=d128k10-p64=0 timeout 300 python tools/wide_bench.py 2>&1 | tail -1
AM_PSTAT64=1 AB_DIM=128 AB_K=10 AB_TAG=d128k10-p64=1 timeout 300 python tools/wide_bench.py 2>&1 | tail -1
AM_PSTAT64=0 AB_DIM=128 AB_DATA=clap AB_K=10 AB_TAG=d128clap-p64=0 timeout 300 python tools/wide_bench.py 2>&1 | tail -1
AM_PSTAT64=1 AB_DIM=128 AB_DATA=clap AB_K=10 AB_TAG=d128clap-p64=1 timeout 300 python tools/wide_bench.py 2>&1 | tail -1
AM_PSTAT64=1 AB_DIM=512 AB_TAG=d512 timeout 300 python tools/wide_bench.py 2>&1 | tail -1
