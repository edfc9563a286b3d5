set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "mnist or link or conv" 2>&1 | tail -5
python bench.py --workload mnist --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "conv|ms_per_step|device-time" | cut -c1-220
ARVAE_CONV64_NO_STAGE=1 python bench.py --workload mnist --breakdown --no-cpu-baseline --no-secondary 2>&1 | grep -E "conv|ms_per_step|device-time" | cut -c1-220
