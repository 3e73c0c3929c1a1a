for lib in libprev_dev.so dev libprev_dev.so dev; do AB_REPS=5 AM_HIP_LIBRARY=$lib timeout 300 python tools/ab_knn.py 2>&1 | tail -1 | sed "s/^/$lib: /" | cut -c1-170; done
for lib in libprev_dev.so dev; do AB_K=10 AB_DATA=unit AB_REPS=4 AM_HIP_LIBRARY=$lib timeout 300 python tools/ab_knn.py 2>&1 | tail -1 | sed "s/^/$lib: /" | cut -c1-170; done
