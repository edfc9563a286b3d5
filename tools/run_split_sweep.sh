#!/bin/bash
# same-box sweep of the workgroup split of a paired backward launch (per cent of the workgroups on the data gradient), kernel-trace
# averages: tools/run_split_sweep.sh [16D|16U|8D|8U|_C1] "44 46 48 50 54 58"
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
which=${1:-16D}
var=ARVAE_PAIR_SPLIT$which
{
for sp in ${2:-44 46 48 50 54 58}; do
  export $var=$sp
  rm -rf /tmp/sw_k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sw_k -o p -- python3 bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs > /dev/null 2>&1
  echo "$var=$sp"
  python3 tools/kstats.py $(find /tmp/sw_k -name '*kernel_stats.csv' | head -1) 50 30 | grep -i "pair_\|kernels,"
done
} > gpurun_out/split_sweep_$which.txt 2>&1
cat gpurun_out/split_sweep_$which.txt
