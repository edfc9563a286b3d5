#!/bin/bash
# Data-parallel evidence of a round on ONE box (a one-GPU box can only run the data-parallel CODE PATH on one rank):
#   1. bench.py plain and --force-dp interleaved three times on this box (same build, same box: the ratio is what counts);
#   2. launch-ordered kernel trace of the forced data-parallel step (library collectives: RCCL on the launch stream);
#   3. the stress loop of the data-parallel MeasureVAE graph-replay path (tools/dp_replay_loop.sh).
#   bash tools/run_dp_check.sh [tag=r5]   -> gpurun_out/dp_check_<tag>.txt, gpurun_out/<tag>_dp_timeline.txt
cd "$(dirname "$0")/.."
tag=${1:-r6}
out=gpurun_out/dp_check_$tag.txt
mkdir -p gpurun_out
: > $out
ms() { python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print('%.4f ms/step  %s' % (d['ms_per_step'], d['config'].get('collectives')))" $1; }
for i in 1 2 3; do
    python3 bench.py --no-cpu-baseline --no-secondary > /tmp/dpc_plain.json 2> /dev/null
    python3 bench.py --no-cpu-baseline --no-secondary --force-dp > /tmp/dpc_dp.json 2> /dev/null
    echo "run $i: plain $(ms /tmp/dpc_plain.json) | forced DP (one rank) $(ms /tmp/dpc_dp.json)" >> $out
done
python3 bench.py --workload measure --no-cpu-baseline > /tmp/dpc_m.json 2> /dev/null
python3 bench.py --workload measure --no-cpu-baseline --force-dp > /tmp/dpc_mdp.json 2> /dev/null
echo "MeasureVAE (whole-model executor, eager): plain $(ms /tmp/dpc_m.json) | forced DP (one rank) $(ms /tmp/dpc_mdp.json)" >> $out
python3 - >> $out <<'P'
import json
for name, f in (('dSprites', '/tmp/dpc_dp.json'), ('MeasureVAE', '/tmp/dpc_mdp.json')):
    d = json.load(open(f))
    print(f'{name}, forced DP: measured cost of the step\'s collectives and the overlap decision (bench.py dp object): {json.dumps(d.get("dp"))}')
P
{
  echo "== forced data-parallel step on ONE rank (bench.py --force-dp, B = 512), launch-ordered kernel medians under rocprofv3 --kernel-trace =="
  echo "-- every collective is an RCCL call of libarvae_hip.so on the launch stream (arvae_comm_*); the library finishes the pass (arvae_image_vae_finish) --"
  BENCH_ARGS=--force-dp bash tools/trace_kernels.sh all
  echo "-- the plain step on the same box --"
  bash tools/trace_kernels.sh all | tail -1
} > gpurun_out/${tag}_dp_timeline.txt 2>&1
bash tools/dp_replay_loop.sh 30 > /dev/null 2>&1
cat gpurun_out/dp_loop.txt >> $out
cat $out
