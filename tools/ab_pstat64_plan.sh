export AM_HIP_LIBRARY=dev AB_REPS=5 AB_DIM=128
for tgt in 4096 8192 16384; do for ms in 16 32 64; do
  AM_KNN_WIDE_WG_TARGET=$tgt AM_KNN_SYM_MAX_SLICES=$ms AB_TAG=p64-tgt$tgt-ms$ms timeout 300 python tools/wide_bench.py 2>&1 | tail -1
done; done
for st in 12 16 32; do
  AM_KNN_SYM_STRIDE=$st AB_TAG=p64-stride$st timeout 300 python tools/wide_bench.py 2>&1 | tail -1
done
AM_PSTAT64=0 AB_TAG=p64off timeout 300 python tools/wide_bench.py 2>&1 | tail -1
AB_TAG=p64on timeout 300 python tools/wide_bench.py 2>&1 | tail -1
python tools/fad_f32_probe.py
