# GPU session r02bx: full suite + fuzz with the moments marks as default
O=gpurun_out/r02bx; mkdir -p $O
timeout -k 10 300 python tools/stress_random.py 200 11000 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "full suite rc=$?"; tail -3 $O/full.log
exit 0
