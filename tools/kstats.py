"""Print the top kernels of a rocprofv3 --kernel-trace --stats CSV (per-step figures when --steps is given)."""
import csv, re, sys
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'{len(rows)} kernels, {tot / 1e3 / steps:.1f} us/step')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = re.sub(r'\(.*', '', r['Name'])[:64]
    print(f"{n:64s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e3 / steps:9.1f} us/step  avg {float(r['AverageNs']) / 1e3:7.1f} us {float(r['Percentage']):5.1f}%")
