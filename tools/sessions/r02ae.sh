# GPU session r02ae: unit-node ranking on the main inverse (new test), then the full suite
O=gpurun_out/r02ae; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "long_cycle or low_entropy or tiny_cycles or wide_path_small" > $O/inv_tests.log 2>&1; echo "inverse tests rc=$?"; tail -5 $O/inv_tests.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "full suite rc=$?"; tail -5 $O/full.log
exit 0
