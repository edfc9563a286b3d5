#!/bin/bash
# the two --pmc passes of tools/profile_round.sh alone: tools/pmc_round.sh <tag>
cd "$(dirname "$0")/.."
tag=${1:-r2}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -o p -- python3 bench.py --steps 3 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 tools/pmc_traffic.py $(find /tmp/pmc_fetch -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_write -name '*counter_collection.csv' | head -1) $out/pmc_traffic.json > $out/pmc_traffic.txt
cat $out/pmc_traffic.txt
