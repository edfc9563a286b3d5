python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for wl in dsprites mnist; do for v in 1 0; do
ARVAE_MIDBLOCK=$v python bench.py --workload $wl --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d["value"]), round(d["ms_per_step"],4))' $wl $v
done; done
