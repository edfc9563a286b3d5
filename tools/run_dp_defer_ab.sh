#!/bin/bash
# round 6: the data-parallel step (one forced rank) with its finishing step carried by the backward pass against a launch of its own
cd "$(dirname "$0")/.."
python -m pytest tests/test_parallel_gpu.py -q -m gpu -k "rccl_ranks_equal or headline_batch or overlapped or bench_under" 2>&1 | tail -4
for i in 1 2 3; do
 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("plain", round(d["ms_per_step"],4))'
 python bench.py --no-cpu-baseline --no-secondary --force-dp 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("forced DP", round(d["ms_per_step"],4))'
 ARVAE_DEFER_FINISH=0 python bench.py --no-cpu-baseline --no-secondary --force-dp 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("forced DP, own finishing launch", round(d["ms_per_step"],4))'
done
