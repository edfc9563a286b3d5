set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline or fused or dsprites or epoch or graph" 2>&1 | tail -3
python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "down32|ms_per_step" | cut -c1-200
ARVAE_DOWN32_REGW=1 python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "down32|ms_per_step" | cut -c1-200
