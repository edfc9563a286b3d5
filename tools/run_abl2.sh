python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "mnist or link or conv" 2>&1 | tail -2
bash tools/ab_libs.sh tools/bin/lib_new2.so tools/bin/lib_new4.so --workload mnist
