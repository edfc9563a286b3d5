#!/bin/bash
# Round-5 SQ counters (north_star: "rocprof-reported ... MFMA utilisation"): one rocprofv3 --pmc pass per workload of
#   SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
# (counters only: no trace domain beside them; the program directly after `--`), summarised per kernel for the kernels with the
# most time in a step:  tools/pmc_sq_round.sh [tag]  ->  gpurun_out/prof_<tag>/<workload>_sq_counters.txt, sq_counters.json
cd "$(dirname "$0")/.."
tag=${1:-r6}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for wl in dsprites mnist measure; do
  rm -rf /tmp/sq_$wl
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/sq_$wl -o p -- python3 bench.py --workload $wl --steps 4 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs > /dev/null 2> /tmp/sq_$wl.err
done
python3 - $out <<'P'
import collections, csv, glob, json, sys
out = sys.argv[1]
summary = {}
for wl in ('dsprites', 'mnist', 'measure'):
    fs = glob.glob(f'/tmp/sq_{wl}/**/*counter_collection.csv', recursive=True)
    if not fs:
        continue
    disp = collections.defaultdict(dict)                  # dispatch -> counters, name, duration
    for r in csv.DictReader(open(fs[0])):
        d = disp[r['Dispatch_Id']]
        d['name'] = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].replace('arvae::', '')
        d['ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        d[r['Counter_Name']] = float(r['Counter_Value'])
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for d in disp.values():
        if 'arvae' not in d['name'] and '_kernel' not in d['name']:
            continue
        f = fam[d['name']]
        f['launches'] += 1
        for k, v in d.items():
            if k != 'name':
                f[k] += v
    tot_ns = sum(f['ns'] for f in fam.values())
    rows = sorted(fam.items(), key=lambda kv: -kv[1]['ns'])
    lines = [f'SQ counters of the {wl} training step (rocprofv3 --pmc, one pass, bench.py --workload {wl} --steps 4 --warmup 2; per launch, mean over the',
             'launches of the pass).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of SIMD-cycles in which the',
             'matrix pipe was busy (GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, 32 per 32x32x16 16-bit MFMA).', '']
    summary[wl] = {}
    for name, f in rows[:12]:
        n = f['launches']
        cyc = f.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        busy = f.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        frac = busy / (1024.0 * cyc) if cyc else 0.0
        rec = {'launches_in_pass': int(n), 'share_of_kernel_time': f['ns'] / tot_ns, 'avg_launch_us_profiled': f['ns'] / n / 1e3,
               'mfma_busy_frac': frac, 'mfma_insts_per_launch': f.get('SQ_INSTS_MFMA', 0) / n, 'valu_insts_per_launch': f.get('SQ_INSTS_VALU', 0) / n,
               'valu_per_mfma': (f.get('SQ_INSTS_VALU', 0) / f['SQ_INSTS_MFMA']) if f.get('SQ_INSTS_MFMA') else None,
               'clock_ghz_profiled': cyc / f['ns'] if f['ns'] else None}
        summary[wl][name] = rec
        lines.append(f"{name[:70]:70s} launches {int(n):4d}  {100 * rec['share_of_kernel_time']:5.1f} % of kernel time  {rec['avg_launch_us_profiled']:8.1f} us  "
                     f"mfma_busy {frac:5.3f}  MFMA {rec['mfma_insts_per_launch']:.3g}  VALU {rec['valu_insts_per_launch']:.3g}"
                     + (f"  VALU/MFMA {rec['valu_per_mfma']:.2f}" if rec['valu_per_mfma'] else ''))
        for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'):
            lines.append(f"    {c:28s} {f.get(c, 0) / n:.4g} per launch")
    open(f'{out}/{wl}_sq_counters.txt', 'w').write('\n'.join(lines) + '\n')
json.dump(summary, open(f'{out}/sq_counters.json', 'w'), indent=1)
P
