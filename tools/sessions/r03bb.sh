# GPU session r03bb: full gpu suite (text 6 GiB test, direct wide mode, release_memory)
O=gpurun_out/r03bb; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -16 $O/pytest_gpu.log
