python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline or conv_up or fused or ragged" 2>&1 | tail -2
bash tools/trace_kernels.sh up32x | tail -8
