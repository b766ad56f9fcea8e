# GPU session r03ap: the child run of test_alternate_paths[BWTS_DENSE=tiles] with its whole output kept
O=gpurun_out/r03ap; mkdir -p $O
BWTS_TEST_CHILD=1 BWTS_TEST_KNOBS=1 BWTS_DENSE=tiles BWTS_TRACE_ERRORS=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q \
  -k "(small or mid_size or deep_repeats or dense_ties or dense_rounds or text_16MiB or reference_unbwts_vectors_through_cabi) and not alternate" > $O/child.log 2>&1
echo "rc=$?"; head -60 $O/child.log
