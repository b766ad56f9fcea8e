# GPU session r03ao: full gpu suite after the wide path's tied list moved to blocks / parts
O=gpurun_out/r03ao; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
