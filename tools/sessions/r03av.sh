# GPU session r03av: full gpu suite + chunk stress after the WIDE chunks
O=gpurun_out/r03av; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
STRESS_BUDGET_S=240 timeout -k 10 400 python tools/stress_chunks.py 400 0 > $O/stress_chunks.txt 2>&1; echo "stress rc=$?"; tail -2 $O/stress_chunks.txt; grep -c " OK" $O/stress_chunks.txt
