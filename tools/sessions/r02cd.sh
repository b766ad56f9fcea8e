# GPU session r02cd: the parity tests with every input sent through the 64-bit paths first (fallback to the main path where they decline)
O=gpurun_out/r02cd; mkdir -p $O
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=16 BWTS_WIDE_BUCKET=65536 BWTS_TEST_CHILD=1 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small or mid_size or kat or golden or reference_unbwts or deep_repeats or dense_ties or dense_rounds or unaligned or tiny or errors or survey" > $O/tests.log 2>&1; echo "rc=$?"; tail -4 $O/tests.log
exit 0
