#!/bin/bash
# Stress loop for the data-parallel MeasureVAE graph-replay path: the one-rank RCCL worker of
# tests/test_parallel_gpu.py::test_measure_data_parallel_step_replays_from_graphs N times in FRESH processes, full output of
# every failure kept.  Round 3's abort (torch's ProcessGroupNCCL watchdog polling an event while the next capture began)
# was intermittent: one green run proves nothing, 30 of 30 on two boxes is the bar.
#   bash tools/dp_replay_loop.sh [N=30] [transport=library]   -> gpurun_out/dp_loop.txt
cd "$(dirname "$0")/.."
N=${1:-30}
export ARVAE_DP_TRANSPORT=${2:-library}
mkdir -p gpurun_out/dp_loop
ok=0
for i in $(seq 1 $N); do
    port=$((29600 + i))
    if timeout 300 python tests/dp_measure_worker.py 0 1 $port /tmp/dp_loop_$i.npz 32 2 > gpurun_out/dp_loop/run_$i.log 2>&1; then
        ok=$((ok + 1)); rm -f gpurun_out/dp_loop/run_$i.log
    else
        echo "run $i failed with $?" >> gpurun_out/dp_loop/failures.txt
    fi
done
echo "dp_replay_loop: $ok / $N passed (transport $ARVAE_DP_TRANSPORT, $(python -c 'import torch; print(torch.cuda.get_device_name(0))' 2>/dev/null), host $(hostname))" | tee gpurun_out/dp_loop.txt
[ $ok -eq $N ]
