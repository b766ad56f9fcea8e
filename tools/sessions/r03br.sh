# GPU session r03br: test_alternate_paths alone, children uncaptured (-s), poison on -- one run to see what a dying child says
O=gpurun_out/r03br; mkdir -p $O
BWTS_TEST_KNOBS=1 BWTS_POISON=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternate_paths" > $O/pytest_alt.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_alt.log
