# GPU session r02bq: inverse of the real-text transform by class, moments vs log
O=gpurun_out/r02bq; mkdir -p $O
for mode in moments log; do echo $mode; BWTS_INV_TRACE=1 BWTS_INV_MARK=$mode timeout -k 10 200 python tools/diag/inv_realtext.py 2>&1 | tail -4; done > $O/out.log 2>&1
cat $O/out.log
exit 0
