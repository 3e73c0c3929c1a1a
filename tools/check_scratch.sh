#!/bin/bash
# Lists every kernel of the library that spills to scratch memory (none should: a spilling tile kernel runs 3-15x slower).
# Usage: tools/check_scratch.sh        (cross-compiles each .hip with the flags of audio-metrics_amd/_build.py plus
#                                       -Rpass-analysis=kernel-resource-usage; a few minutes)
cd "$(dirname "$0")/../audio-metrics_amd/csrc" || exit 1
FLAGS=$(python3 -c "
import importlib.util, os
spec = importlib.util.spec_from_file_location('b', os.path.join('..', '_build.py')); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(' '.join(b.HIPCC_FLAGS))")
rc=0
for f in *.hip; do
  out=$(hipcc $FLAGS --cuda-device-only -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
        awk '/Function Name:/{name=$0} /ScratchSize \[bytes\/lane\]: [1-9]/{print name; print $0}')
  if [ -n "$out" ]; then echo "== $f"; echo "$out" | sed 's/.*remark: //'; rc=1; fi
done
[ $rc -eq 0 ] && echo "no kernel uses scratch memory"
exit $rc
