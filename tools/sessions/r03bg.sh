# GPU session r03bg: constant input at 2^32 (identity, both ways), and the tests around constant / low-entropy inputs
O=gpurun_out/r03bg; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "constant or low_entropy or attempts or small or known or structured or tiny_cycles" > $O/pytest.log 2>&1; echo "rc=$?"; tail -4 $O/pytest.log
