set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline or fused or epoch or ragged" 2>&1 | tail -2
bash tools/trace_kernels.sh heads | tail -4
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-150
ARVAE_NO_HEADS_NEXT=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-150
