# GPU session r03bq: the whole gpu suite with every arena and side block poisoned before use (BWTS_POISON=1)
O=gpurun_out/r03bq; mkdir -p $O
BWTS_TEST_KNOBS=1 BWTS_POISON=1 timeout -k 10 1150 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity.py::test_stray_knobs_are_ignored_without_the_gate --durations=5 > $O/pytest_gpu_poison.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest_gpu_poison.log
