# GPU session r03bf: every stress tool on the final build of round 3
O=gpurun_out/r03bf; mkdir -p $O
timeout -k 10 400 python tools/stress_random.py 300 0 > $O/stress_random.txt 2>&1; echo "random rc=$? $(tail -1 $O/stress_random.txt)"
timeout -k 10 500 python tools/stress_dense.py 40 0 > $O/stress_dense.txt 2>&1; echo "dense rc=$? $(tail -1 $O/stress_dense.txt)"
STRESS_BUDGET_S=200 timeout -k 10 400 python tools/stress_chunks.py 400 400 > $O/stress_chunks.txt 2>&1; echo "chunks rc=$? $(tail -1 $O/stress_chunks.txt)"
STRESS_BUDGET_S=200 timeout -k 10 400 python tools/stress_wide_parts.py 400 400 > $O/stress_wide_parts.txt 2>&1; echo "wide parts rc=$? $(tail -1 $O/stress_wide_parts.txt)"
timeout -k 10 400 python tools/stress_unreached.py 24 > $O/stress_unreached.txt 2>&1; echo "unreached rc=$? $(tail -1 $O/stress_unreached.txt)"
