#!/bin/bash
# SQ counters of one workload's kernels: tools/pmc_sq.sh <workload> <kernel substring>   (outputs gpurun_out/pmc_sq_<workload>.txt)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
wl=${1:-mnist}; pat=${2:-conv_wgrad_pairs}
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d /tmp/pmc_sq -o p -- python3 bench.py --workload $wl --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pmc_sq2 -o p -- python3 bench.py --workload $wl --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-secondary --no-graphs > /dev/null 2>&1
PAT="$pat" python3 - <<'P' > gpurun_out/pmc_sq_$wl.txt
import csv, glob, os, collections
pat = os.environ['PAT']
for d in ('/tmp/pmc_sq', '/tmp/pmc_sq2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('no counters in', d); continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        if pat in r['Kernel_Name']:
            k = r['Kernel_Name'].split('(')[0][-60:] + ' grid ' + r.get('Grid_Size', '?')
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'SQ_LDS_IDX_ACTIVE'): cnt[k] += 1
    for k, v in tot.items():
        n = max(cnt[k], 1)
        print(k, 'launches', n)
        for c, x in sorted(v.items()):
            print('   %-28s %.4g per launch' % (c, x / n))
P
