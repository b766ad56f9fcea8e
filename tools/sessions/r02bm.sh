# GPU session r02bm: 64-bit inverse with per-range moments -- forced small tests, fuzz, 12 GiB timing
O=gpurun_out/r02bm; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide" > $O/wide_tests.log 2>&1; echo "wide tests rc=$?"; tail -3 $O/wide_tests.log
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=12 BWTS_WIDE_BUCKET=4096 timeout -k 10 300 python tools/stress_random.py 160 9000 > $O/stress_wide.log 2>&1; echo "wide fuzz rc=$?"; tail -1 $O/stress_wide.log
timeout -k 10 400 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; grep "inverse\|round trip" $O/wide12.log | cut -c1-260
exit 0
