set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conv_down_vs or conv_up_vs or image_step or baseline_batch or c1" 2>&1 | tail -3
for w in 4 8 16; do echo waves $w; ARVAE_C1_DOWN_WAVES=$w python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "down_c1|ms_per_step" | cut -c1-200; done
echo tiled; ARVAE_C1_DOWN_TILED=1 python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "down_c1|ms_per_step" | cut -c1-200
