#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (outputs under gpurun_out/prof_<tag>/):
#   tools/profile_round.sh r2
# 1. kernel-trace stats of the default bench command (dSprites) and of the two secondary workloads,
# 2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass) for the HBM traffic of the dSprites kernels.
cd "$(dirname "$0")/.."
tag=${1:-r6}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for wl in dsprites mnist measure; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python3 bench.py --workload $wl --steps 50 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs > $out/${wl}_bench.json 2> /tmp/prof_$wl.err
  cp $(find /tmp/prof_$wl -name '*kernel_stats.csv' | head -1) $out/${wl}_kernel_stats.csv
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o p -- python3 bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 tools/pmc_traffic.py $(find /tmp/pmc_fetch -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_write -name '*counter_collection.csv' | head -1) $out/pmc_traffic.json > $out/pmc_traffic.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --breakdown --no-cpu-baseline --no-secondary > /dev/null 2> $out/breakdown.txt
bash tools/pmc_sq_round.sh $tag
for wl in dsprites mnist measure; do bash tools/trace_kernels.sh all WORKLOAD=$wl > $out/${wl}_launch_trace.txt 2>&1; done
