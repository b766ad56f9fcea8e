# GPU session r02bb: new repeated-material tests, on the default path and under every alternate path
O=gpurun_out/r02bb; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense_rounds or alternate" > $O/tests.log 2>&1; echo "rc=$?"; tail -4 $O/tests.log
exit 0
