#!/bin/bash
# round 6: does the second read of g in pair(down32<16> + wgrad32<16>) come from HBM?  At a 50 % split both halves give every
# workgroup the same two images (8 tiles / 16 stream steps) and workgroup w of the data gradient shares its XCD (w % 8) with
# workgroup w of the weight gradient: the later reader can hit that XCD's L2.  At the default 48 % (122 + 128 workgroups, 9 tiles
# against 16 steps) the ranges and the XCDs drift apart.  Time (bench + kernel trace) and HBM bytes (two --pmc passes) both ways.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
{
for rep in 1 2 3; do
  for sp in ${SPLITS:-48 50}; do
    export ARVAE_PAIR_SPLIT16D=$sp
    echo "split $sp  $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  done
done
for sp in ${SPLITS:-48 50}; do
  export ARVAE_PAIR_SPLIT16D=$sp
  rm -rf /tmp/l2_k /tmp/l2_f /tmp/l2_w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/l2_k -o p -- python3 bench.py --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs > /dev/null 2>&1
  echo "split $sp kernel trace:"
  python3 tools/kstats.py $(find /tmp/l2_k -name '*kernel_stats.csv' | head -1) 50 30 | grep -i "pair_down_wgrad\|kernels,"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/l2_f -o p -- python3 bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/l2_w -o p -- python3 bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  echo "split $sp HBM bytes:"
  python3 tools/pmc_traffic.py $(find /tmp/l2_f -name '*counter_collection.csv' | head -1) $(find /tmp/l2_w -name '*counter_collection.csv' | head -1) /tmp/l2_traffic.json | grep "pair(\|all kernels"
done
} > gpurun_out/l2_align_ab.txt 2>&1
cat gpurun_out/l2_align_ab.txt
