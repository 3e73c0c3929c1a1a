# whole k-NN entry (sample pass + sweep + verification + selection) by sample stride at the widths of the 256-thread engine
export AM_HIP_LIBRARY=dev AB_REPS=9
for d in 64 128; do for st in 8 12 16 24 32; do
  echo -n "D=$d stride $st: "; AM_KNN_SYM_STRIDE=$st AB_DIM=$d timeout 300 python tools/ab_knn.py 2>&1 | tail -1
done; done
for d in 64 128; do echo -n "D=$d shipped: "; AB_DIM=$d timeout 300 python tools/ab_knn.py 2>&1 | tail -1; done
