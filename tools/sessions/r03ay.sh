# GPU session r03ay: the new test of groups of hundreds and thousands
O=gpurun_out/r03ay; mkdir -p $O
BWTS_ROUND_TRACE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "chunk_rounds" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "chunks\] (list|groups)|passed|failed" $O/pytest.log | head -20
