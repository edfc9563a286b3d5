#!/bin/bash
# tools/trace_kernels.sh <substring> [env assignments...]: launch-ordered durations of the matching kernels in one training
# step of the default bench (rocprofv3 kernel trace)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
pat=$1; shift
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/tr_k
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_k -o p -- python3 bench.py --workload ${WORKLOAD:-dsprites} --steps ${STEPS:-30} --warmup 5 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs ${BENCH_ARGS:-} > /dev/null 2> /tmp/tr_k.err
PAT="$pat" python3 - <<'P'
import csv, glob, os, collections
f = glob.glob('/tmp/tr_k/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# one step = the launches between two adam kernels; take the last complete one and the medians over all steps
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
pat = os.environ['PAT']
def dur(r): return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
steps = [rows[a + 1:b + 1] for a, b in zip(idx[:-1], idx[1:])]
steps = [s for s in steps if len(s) == len(steps[-1])]
med = []
for k in range(len(steps[-1])):
    v = sorted(dur(s[k]) for s in steps)
    med.append(v[len(v) // 2])
tot = 0.0
for k, r in enumerate(steps[-1]):
    if pat in r['Kernel_Name'] or pat == 'all':
        gap = (int(r['Start_Timestamp']) - int(steps[-1][k - 1]['End_Timestamp'])) / 1e3 if k else 0.0
        print('%3d %-60s grid %7s wg %4s  median %6.1f us  (gap before %5.1f)' % (k, r['Kernel_Name'][:60], r['Grid_Size_X'], r['Workgroup_Size_X'], med[k], gap))
        tot += med[k]
print('sum of medians %.1f us over %d launches; step = %d launches, sum of all medians %.1f us' % (tot, sum(1 for r in steps[-1] if pat in r['Kernel_Name'] or pat == 'all'), len(steps[-1]), sum(med)))
P
