#!/bin/bash
# A/B of the sample strides of the two f16 filters (dev build knobs): k-NN bound sample (AM_KNN_SYM_STRIDE) and witness sample
# of the membership filter (AM_FAST_PRE_STRIDE); prints tools/wide_bench.py's line plus the sample kernels' times are in the
# total of tools/size_sweep-style evaluate timing below.
export AM_HIP_LIBRARY=dev
for ks in 16 24 32 48; do
  for cs in 16 32; do
    echo -n "knn_stride=$ks cross_stride=$cs: "
    AM_KNN_SYM_STRIDE=$ks AM_FAST_PRE_STRIDE=$cs AB_REPS=4 python tools/ab_evaluate.py 2>&1 | tail -1
  done
done
