# GPU session r02cc: wider validation at HEAD -- fuzz on new seeds (default paths, 64-bit paths), repeated-material stress
O=gpurun_out/r02cc; mkdir -p $O
timeout -k 10 500 python tools/stress_random.py 700 20000 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=12 BWTS_WIDE_BUCKET=4096 timeout -k 10 500 python tools/stress_random.py 250 30000 > $O/stress_wide.log 2>&1; echo "wide fuzz rc=$?"; tail -1 $O/stress_wide.log
timeout -k 10 600 python tools/stress_dense.py 72 100 > $O/stress_dense.log 2>&1; echo "dense rc=$?"; tail -1 $O/stress_dense.log
exit 0
