export AM_HIP_LIBRARY=dev AB_REPS=5
for d in 512 128; do for g in 8 4 16 32 8; do
  AM_WIDE_GROUP_ROWS=$g AB_DIM=$d AB_TAG=d$d-grp$g timeout 300 python tools/wide_bench.py 2>&1 | tail -1
done; done
for g in 8 16 32; do AM_WIDE_GROUP_ROWS=$g AB_DIM=512 AB_DATA=clap AB_K=10 AB_TAG=clap-grp$g timeout 300 python tools/wide_bench.py 2>&1 | tail -1; done
for tgt in 3072 6144 12288; do AM_WIDE_WG_TARGET=$tgt AM_WIDE_GROUP_ROWS=32 AB_DIM=512 AB_TAG=grp32-tgt$tgt timeout 300 python tools/wide_bench.py 2>&1 | tail -1; done
