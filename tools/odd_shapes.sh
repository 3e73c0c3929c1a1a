#!/bin/bash
# Filter paths against the exact f32 kernels (A/B build with AM_KNN_FAST=0 AM_PRDC_FAST=0) at shapes the fuzzers do not draw:
# candidate sets smaller than a tile, reference sets smaller than a tile, embedding widths 768 ... 5000.
run() {  # rows rows2 dim k data
  for tool in ab_knn.py ab_cross.py; do
    a=$(AM_HIP_LIBRARY=dev AM_KNN_FAST=0 AM_PRDC_FAST=0 AB_ROWS=$1 AB_ROWS2=$2 AB_DIM=$3 AB_K=$4 AB_DATA=$5 AB_REPS=1 AB_WANT_MIN=1 python tools/$tool 2>&1 | grep -o "sha1 [0-9a-f]*" | tail -1)
    b=$(AB_ROWS=$1 AB_ROWS2=$2 AB_DIM=$3 AB_K=$4 AB_DATA=$5 AB_REPS=1 AB_WANT_MIN=1 python tools/$tool 2>&1 | grep -o "sha1 [0-9a-f]*" | tail -1)
    [ "$a" == "$b" ] && [ -n "$a" ] && r=ok || r="MISMATCH $a / $b"
    echo "rows=$1/$2 dim=$3 k=$4 data=$5 $tool: $r"
  done
}
run 100000 170 512 5 randn
run 170 100000 512 5 randn
run 70000 257 64 10 clustered
run 100000 255 128 1 unit
run 65536 256 768 5 randn
run 40000 40000 1024 5 lowrank
run 20000 20000 2048 3 randn
run 33000 1000 3000 5 scales
run 17000 17000 4096 2 randn
run 9000 9000 5000 2 randn
