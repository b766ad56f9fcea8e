# GPU session r02bv: the two new alternate paths
O=gpurun_out/r02bv; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternate and (FUSED or SYMS)" > $O/tests.log 2>&1; echo "rc=$?"; tail -3 $O/tests.log
exit 0
