# where the time of the two filter kernels is: AM_WIDE_DBG of the A/B build (2 = epilogues compiled out, 16 = fast path and
# first gate only, 32 = everything but the queue stores, 0 = the whole kernel); results are meaningless with a knob set
export AM_HIP_LIBRARY=dev AB_REPS=5
for d in 512 128 64; do for dbg in 0 32 16 2; do
  AM_WIDE_DBG=$dbg AB_DIM=$d AB_TAG=d$d-dbg$dbg timeout 300 python tools/wide_bench.py 2>&1 | tail -1
done; done
