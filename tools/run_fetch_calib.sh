#!/bin/bash
# round 6: FETCH_SIZE / WRITE_SIZE against known byte counts (tools/probes/fetch_calib.hip; build it first:
#   hipcc --offload-arch=gfx950 -O3 -o tools/bin/fetch_calib tools/probes/fetch_calib.hip)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
{
rm -rf /tmp/fc_f /tmp/fc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fc_f -o p -- ./tools/bin/fetch_calib
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/fc_w -o p -- ./tools/bin/fetch_calib > /dev/null
python3 - <<'PY'
import csv, glob, collections
known = {'rd_global16': 256 << 20, 'rd_buffer16': 256 << 20, 'rd_pixels': (256 << 20) // 40960 * 40960, 'rd_global4': 256 << 20,
         'wr_global16': 256 << 20, 'rd_twice_l2': 32 << 20}
for d, c in (('/tmp/fc_f', 'FETCH_SIZE'), ('/tmp/fc_w', 'WRITE_SIZE')):
    path = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    vals = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] == c:
            vals[row['Kernel_Name'].split('(')[0]].append(float(row['Counter_Value']))
    for k, v in vals.items():
        b = known.get(k)
        if b:
            print(f"{c:10s} {k:12s} raw {[round(x * 1024 / 1e6, 1) for x in v]} MB for {b / 1e6:.1f} MB moved once  (raw / known = {v[-1] * 1024 / b:.3f})")
PY
} > gpurun_out/fetch_calib.txt 2>&1
cat gpurun_out/fetch_calib.txt
