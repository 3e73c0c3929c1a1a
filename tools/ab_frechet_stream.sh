export AM_HIP_LIBRARY=dev
for round in 1 2 3; do for place in 0 1 2; do
  AM_EVAL_FAD_PLACE=$place AB_TAG=r$round timeout 300 python tools/ab_frechet_stream.py 2>&1 | tail -1
done; done
for place in 0 1 2; do AM_EVAL_FAD_PLACE=$place AB_DIM=128 AB_TAG=d128 timeout 300 python tools/ab_frechet_stream.py 2>&1 | tail -1; done
