set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
python bench.py --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "dense|heads|ms_per_step" | cut -c1-200
