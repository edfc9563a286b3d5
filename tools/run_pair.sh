set -e
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline or fused or epoch or ragged or latent_block" 2>&1 | tail -2
bash tools/trace_kernels.sh pair_c1 | tail -4
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-150
ARVAE_NO_PAIR_C1=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-150
done
