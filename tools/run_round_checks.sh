#!/bin/bash
# The round's GPU evidence in one call (run through gpurun from the repo root):  tools/run_round_checks.sh <tag>
#   1. the whole -m gpu suite; 2. bench.py (default workload, all variants) -> bench_n1.json; 3. the driver's multi-rank
#   launch at the FULL size with 2 / 4 / 8 ranks on one GPU over gloo (plain `python bench.py --gpus N`: bench.py starts its
#   own ranks) -> multi_rank_full_size_one_gpu.txt; 4. bench.py --config e2e --pairs 10000 (configs[4] at its stated size).
TAG=${1:-r4}
OUT=gpurun_out/checks_${TAG}
mkdir -p "$OUT"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > "$OUT/pytest_gpu.txt"
timeout 900 python bench.py --steps 20 --warmup 3 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
{
echo "# python bench.py --gpus N (plain: bench.py starts its own ranks through torch.distributed.run) at the FULL bench size"
echo "# (2 x 100000 x 512, k = 5), every rank on cuda:0 over gloo (AM_BENCH_DEVICE=0 AM_BENCH_BACKEND=gloo: a 1-GPU box cannot host"
echo "# N RCCL ranks).  Times are meaningless (N processes share one GPU, gloo stages through the host); the N-rank flow must"
echo "# reproduce the 1-rank result and the fixture."
echo "# ranks n_ranks_seen  result  result_check.ok"
for n in 8 4 2; do
  AM_BENCH_DEVICE=0 AM_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus $n --steps 1 --warmup 1 --no-cpu-baseline --no-variants 2> "$OUT/bench_n$n.err" | python3 -c "
import json, sys
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['result']
        print(d['n_gpus'], d['n_ranks_seen'], ' '.join(f'{k} {v!r}' for k, v in r.items()), d['result_check']['ok'],
              '| python_schedule_ms', round(d.get('python_schedule_ms') or 0, 1), 'c_entry_ms', round(d.get('c_entry_ms') or 0, 1), '|', d.get('schedules_check'))
"
done
python3 -c "
import json
d = json.load(open('$OUT/bench_n1.json')); r = d['result']
print(d['n_gpus'], d['n_ranks_seen'], ' '.join(f'{k} {v!r}' for k, v in r.items()), d['result_check']['ok'], '  ($OUT/bench_n1.json:', round(d['ms_per_step'], 2), 'ms per step)')
"
} > "$OUT/multi_rank_full_size_one_gpu.txt" 2>&1
timeout 1200 python bench.py --config e2e --pairs 10000 > "$OUT/bench_e2e_10k_n1.json" 2> "$OUT/bench_e2e_10k_n1.err"
tail -3 "$OUT/pytest_gpu.txt"; cat "$OUT/multi_rank_full_size_one_gpu.txt"; head -c 600 "$OUT/bench_e2e_10k_n1.json"
