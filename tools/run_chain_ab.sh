#!/bin/bash
# round 6: conv2 + conv3 of the dSprites forward pass as one launch (conv32.hip chain_down_kernel) against two launches
# (ARVAE_NO_DOWN_CHAIN, diagnostic library both ways)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
{
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline_batch_512 or full_batch or ragged or three_steps or headline or deferred" 2>&1 | tail -3
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2 3; do
  echo "A (chained)      $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "B (two launches) $(ARVAE_NO_DOWN_CHAIN=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done
for v in A B; do
  [ $v = B ] && export ARVAE_NO_DOWN_CHAIN=1
  rm -rf /tmp/ch_k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ch_k -o p -- python3 bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs > /dev/null 2>&1
  echo "$v kernel trace:"
  python3 tools/kstats.py $(find /tmp/ch_k -name '*kernel_stats.csv' | head -1) 50 30 | grep -i "chain_down\|down32p\|kernels,"
done
} > gpurun_out/chain_ab.txt 2>&1
cat gpurun_out/chain_ab.txt
