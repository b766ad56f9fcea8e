# GPU session r03al: stress of the chunked rounds against the oracle (tools/stress_chunks.py), then the larger repeated-material stress
O=gpurun_out/r03al; mkdir -p $O
STRESS_BUDGET_S=300 timeout -k 10 500 python tools/stress_chunks.py 400 0 > $O/stress_chunks.txt 2>&1; echo "chunks rc=$?"; grep -c OK $O/stress_chunks.txt; grep -v " OK$" $O/stress_chunks.txt | tail -8
timeout -k 10 400 python tools/stress_dense.py 24 500 > $O/stress_dense.txt 2>&1; echo "dense rc=$?"; tail -3 $O/stress_dense.txt
