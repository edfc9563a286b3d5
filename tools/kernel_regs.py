#!/usr/bin/env python3
"""tools/kernel_regs.py <file.hip ...>: registers, spills, scratch and occupancy of every kernel in the given sources
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for src in sys.argv[1:]:
    path = src if os.path.exists(src) else os.path.join(root, 'ar-vae_amd', 'csrc', src)
    out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(root, 'include'),
                          '-c', path, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = subprocess.run(['/usr/bin/c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r'\(.*', '', cur).replace('void arvae::', '').replace('arvae::', '')
            rows[cur] = {}
            continue
        m = re.search(r'remark: ([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for k, r in sorted(rows.items()):
        print('%-44s vgpr %3d agpr %3d spill %3d scratch %4d occ %d' % (k[:44], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('VGPRs Spill', -1),
                                                                       r.get('ScratchSize', -1), r.get('Occupancy', -1)))
