#!/bin/bash
# round 5: two processes at B = 512 on one device under variants of the clustered latent block's placement (diagnostic library):
#   tools/run_shared_probe.sh "<VAR=val ...>" [reps]
cd "$(dirname "$0")/.."
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
for kv in $1; do export "$kv"; done
for rep in $(seq 1 ${2:-8}); do
  d=$(mktemp -d); mkdir $d/sync
  python tests/shared_device_worker.py $d/a.npz 200 $d/sync 0 2> $d/a.err & pa=$!
  python tests/shared_device_worker.py $d/b.npz 200 $d/sync 0 2> $d/b.err & pb=$!
  while [ $(ls $d/sync | grep -c ready_) -lt 2 ]; do sleep 0.05; done
  touch $d/sync/go
  wait $pa; ra=$?; wait $pb; rb=$?
  python - $d "$1" $rep $ra $rb <<'P'
import sys, numpy as np, os
d, tag, rep, ra, rb = sys.argv[1:]
out = []
for n, rc in (('a', ra), ('b', rb)):
    if rc == '0':
        r = np.load(f'{d}/{n}.npz'); out.append(f"{n}: ok {float(r['seconds']):.3f}s same={bool(r['same'])}")
    else:
        err = open(f'{d}/{n}.err').read().strip().splitlines()[-1]; i = err.find('status'); out.append(f'{n}: rc {rc} {err[i:i + 40]}'); print(open(f'{d}/{n}.err').read()[:6000])
print(f'[{tag}] rep {rep}: ' + ' | '.join(out))
P
done
