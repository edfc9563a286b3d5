set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv_down_vs or conv_up_vs or wgrad or image_step or baseline_batch or c1" 2>&1 | tail -3
for g in 256 512 1024; do echo grid $g; ARVAE_C1_WGRAD_GRID=$g python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "wgrad_c1|slab|ms_per_step" | cut -c1-200; done
echo tiled; ARVAE_C1_WGRAD_TILED=1 python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "wgrad_c1|slab|ms_per_step" | cut -c1-200
