#!/bin/bash
# rocprofv3 kernel stats of tools/time_link.py:  tools/prof_link.sh wgrad 16   (run on the GPU box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=/tmp/prof_$1_$2_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/time_link.py "$@" > $out.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then tail -5 $out.log; find $out | head; else python3 tools/kstats.py $f 220 6; fi
