set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline or conv_up" 2>&1 | tail -2
python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "up32|ms_per_step" | cut -c1-200
