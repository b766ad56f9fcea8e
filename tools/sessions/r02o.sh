# GPU session r02o: randomized parity runs (forward, inverse, inverse of arbitrary bytes) on the default paths and on the forced-wide paths
set -o pipefail
O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 500 python tools/stress_random.py 160 1000 > $O/stress_default.log 2>&1; echo "default rc=$?"; tail -3 $O/stress_default.log
BWTS_DENSE_RUNS=1 timeout -k 10 300 python tools/stress_random.py 80 5000 > $O/stress_runs.log 2>&1; echo "runs rc=$?"; tail -2 $O/stress_runs.log
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=12 BWTS_WIDE_BUCKET=4096 timeout -k 10 300 python tools/stress_random.py 60 9000 > $O/stress_wide.log 2>&1; echo "wide rc=$?"; tail -2 $O/stress_wide.log
