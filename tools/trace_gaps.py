"""Summarise a rocprofv3 --kernel-trace CSV: busy time vs idle gaps between consecutive kernels, per training step."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# one step = from one adam_step kernel to the next
adam = [i for i, n in enumerate(names) if 'adam' in n.lower()]
if len(adam) < 12:
    sys.exit('too few steps in trace')
lo, hi = adam[-11], adam[-1]                      # last 10 steps
seg = rows[lo + 1:hi + 1]
steps = 10
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(seg[:-1], seg[1:])]
print(f'kernels/step {len(seg) / steps:.1f}  span/step {(t1 - t0) / steps / 1e3:.1f} us  busy/step {busy / steps / 1e3:.1f} us  '
      f'gap/step {sum(gaps) / steps / 1e3:.1f} us  mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us  overlapped {sum(1 for g in gaps if g < 0)}')
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r['Kernel_Name'].split('(')[0][:70]
    per[k][0] += 1
    per[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f'  {k:70s} {c / steps:5.1f}/step {t / steps / 1e3:8.1f} us/step  {t / c / 1e3:7.1f} us each')
if len(sys.argv) > 2:
    print('--- last step, in order')
    one = rows[adam[-2] + 1:adam[-1] + 1]
    for r in one:
        print(f"  {(int(r['Start_Timestamp']) - int(one[0]['Start_Timestamp'])) / 1e3:8.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} us  {r['Kernel_Name'][:90]}  grid {r.get('Grid_Size_X', '?')} wg {r.get('Workgroup_Size_X', '?')}")
