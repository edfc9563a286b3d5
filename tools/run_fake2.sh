cd /root/repo; mkdir -p gpurun_out
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2; do
  echo "A (default)  $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "B (fake2)  $(ARVAE_LIB=tools/bin/lib_fake2.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/fake_bench.txt 2>&1
bash tools/trace_kernels.sh all > gpurun_out/fake_trace_a.txt 2>&1
bash tools/trace_kernels.sh all ARVAE_LIB=tools/bin/lib_fake2.so > gpurun_out/fake_trace_b.txt 2>&1
