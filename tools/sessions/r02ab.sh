# GPU session r02ab: wide inverse with the unit-node ranking of long splitter-free cycles (tests, fuzz in one context)
O=gpurun_out/r02ab; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_path_small" > $O/wide_tests.log 2>&1; echo "wide tests rc=$?"; tail -4 $O/wide_tests.log
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=12 BWTS_WIDE_BUCKET=4096 timeout -k 10 500 python tools/stress_random.py 160 9000 > $O/stress_wide.log 2>&1; echo "wide fuzz rc=$?"; tail -4 $O/stress_wide.log
exit 0
