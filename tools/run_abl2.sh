python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline or conv_down or fused or ragged" 2>&1 | tail -2
bash tools/trace_kernels.sh down32k | tail -6
bash tools/ab_libs.sh tools/bin/lib_new.so tools/bin/lib_new2.so
