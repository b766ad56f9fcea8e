# GPU session r02bo: why the text workload's inverse falls back to the log
O=gpurun_out/r02bo; mkdir -p $O
BWTS_INV_TRACE=1 timeout -k 10 300 python bench.py --workload text --steps 1 --warmup 0 --inverse-steps 1 --breakdown-steps 0 --no-cpu-baseline --no-e2e > $O/bench_text.log 2>&1; grep "\[inverse\]" $O/bench_text.log | head
BWTS_INV_TRACE=1 timeout -k 10 300 python tools/check_realtext.py 26 > $O/realtext.log 2>&1; grep "\[inverse\]" $O/realtext.log | head -4
exit 0
