# GPU session r03bm: the pinned-host in/out test
O=gpurun_out/r03bm; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned_host or host_path" > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
