# GPU session r02be: 64-bit paths with the default bucket size -- the suite's wide tests (12 GiB round trip among them)
O=gpurun_out/r02be; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide" > $O/wide_tests.log 2>&1; echo "wide tests rc=$?"; tail -3 $O/wide_tests.log
exit 0
