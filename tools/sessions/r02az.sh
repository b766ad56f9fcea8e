# GPU session r02az: full suite + fuzz at HEAD (carried byte in the 64-bit path, unrolled counting loop)
O=gpurun_out/r02az; mkdir -p $O
timeout -k 10 300 python tools/stress_random.py 200 8000 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "full suite rc=$?"; tail -3 $O/full.log
exit 0
