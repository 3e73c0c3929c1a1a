#!/bin/bash
# Lists every kernel of the library that spills to scratch memory (none should: a spilling tile kernel runs 3-15x slower).
# Usage: tools/check_scratch.sh        (cross-compiles each .hip with -Rpass-analysis=kernel-resource-usage, a few minutes)
cd "$(dirname "$0")/../audio-metrics_amd/csrc" || exit 1
rc=0
for f in *.hip; do
  out=$(hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
        awk '/Function Name:/{name=$0} /ScratchSize \[bytes\/lane\]: [1-9]/{print name; print $0}')
  if [ -n "$out" ]; then echo "== $f"; echo "$out" | sed 's/.*remark: //'; rc=1; fi
done
[ $rc -eq 0 ] && echo "no kernel uses scratch memory"
exit $rc
