#!/bin/bash
# Round 6: the same check as odd_shapes.sh - filter paths against the exact f32 kernels (A/B build with AM_KNN_FAST=0
# AM_PRDC_FAST=0) - at the widths of the 256-thread stationary engine (rows of up to two 64-element slabs) and around its
# boundary, incl. candidate / reference sets smaller than a tile and k = 1 / 10.
run() {  # rows rows2 dim k data
  for tool in ab_knn.py ab_cross.py; do
    a=$(AM_HIP_LIBRARY=dev AM_KNN_FAST=0 AM_PRDC_FAST=0 AB_ROWS=$1 AB_ROWS2=$2 AB_DIM=$3 AB_K=$4 AB_DATA=$5 AB_REPS=1 AB_WANT_MIN=1 python tools/$tool 2>&1 | grep -o "sha1 [0-9a-f]*" | tail -1)
    b=$(AB_ROWS=$1 AB_ROWS2=$2 AB_DIM=$3 AB_K=$4 AB_DATA=$5 AB_REPS=1 AB_WANT_MIN=1 python tools/$tool 2>&1 | grep -o "sha1 [0-9a-f]*" | tail -1)
    [ "$a" == "$b" ] && [ -n "$a" ] && r=ok || r="MISMATCH $a / $b"
    echo "rows=$1/$2 dim=$3 k=$4 data=$5 $tool: $r"
  done
}
run 100000 170 128 5 randn
run 170 100000 128 5 randn
run 70000 70000 3 5 randn
run 50000 257 17 10 unit
run 65536 65536 33 1 randn
run 40000 40000 64 10 clustered
run 33000 1000 65 5 scales
run 100000 255 100 1 unit
run 47001 47001 127 3 dups
run 81000 81000 128 10 silence
run 20000 20000 129 5 randn
run 150000 150000 64 5 randn
