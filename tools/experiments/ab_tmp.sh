mkdir -p gpurun_out/r5
O=gpurun_out/r5/wide_bench_accinit_ab.txt; : > $O
for rep in 1 2; do
for d in 512 128; do
for v in libam_base.so libam_late.so dev; do
AB_TAG=$v-d$d AB_DIM=$d AM_HIP_LIBRARY=$v timeout 300 python tools/wide_bench.py 2>&1 | grep -v amdgpu.ids >> $O
done; done; done
cat $O
