#!/bin/bash
# per-job durations of the batched Linear weight gradient (one launch per job) from a kernel trace
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export ARVAE_DENSE_BATCH_SPLIT=1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_dw -o p -- python3 bench.py --steps 30 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-secondary > /dev/null 2> /tmp/tr_dw.err
python3 - <<'P'
import csv, glob, collections
f = glob.glob('/tmp/tr_dw/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'dense_wgrad_batch' in r['Kernel_Name'] or 'slab_reduce' in r['Kernel_Name']:
        g = int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))
        d[(r['Kernel_Name'][:40], g)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = [r for r in csv.DictReader(open(f)) if 'dense_wgrad_batch' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = len(rows) // 35
print('launch order within the last step: (grid, us)', [(int(r['Grid_Size_X']) // 512, round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1)) for r in rows[-per:]])
print('one step earlier:', [(int(r['Grid_Size_X']) // 512, round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1)) for r in rows[-2 * per:-per]])
for k, v in sorted(d.items()):
    v = sorted(v)
    print(k, 'n', len(v), 'median %.1f us' % (v[len(v) // 2] / 1e3), 'min %.1f' % (v[0] / 1e3))
P
