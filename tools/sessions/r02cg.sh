# GPU session r02cg: inverse with chosen numbers of unreached elements (arithmetic / search / fallback), main and 64-bit paths
O=gpurun_out/r02cg; mkdir -p $O
BWTS_INV_TRACE=1 timeout -k 10 500 python tools/stress_unreached.py 22 > $O/main.log 2>&1; echo "main rc=$?"; grep "^m \|^bad" $O/main.log | tail -26; grep -c "fallback flag [1-9]" $O/main.log; grep -c "ranges searched [1-9]" $O/main.log
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=16 BWTS_WIDE_BUCKET=65536 timeout -k 10 500 python tools/stress_unreached.py 20 > $O/wide.log 2>&1; echo "wide rc=$?"; tail -1 $O/wide.log
exit 0
