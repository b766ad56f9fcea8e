# GPU session r02ba: stress of the group-local rounds (large groups, runs, nested copies) against the oracle
O=gpurun_out/r02ba; mkdir -p $O
BWTS_ROUND_TRACE=1 timeout -k 10 900 python tools/stress_dense.py 48 0 > $O/stress_dense.log 2>&1; echo "rc=$?"; grep -c "in larger groups [1-9]" $O/stress_dense.log; grep "^seed\|^cases" $O/stress_dense.log | tail -50
exit 0
