# GPU session r02ak: per-round list sizes and the share of larger groups on the three texts
O=gpurun_out/r02ak; mkdir -p $O
BWTS_ROUND_TRACE=1 timeout -k 10 300 python tools/time_realtext.py 26 1 > $O/realtext.log 2>&1; echo "rc=$?"; grep "rounds\]" $O/realtext.log | head -12
BWTS_ROUND_TRACE=1 timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "rc=$?"; grep "rounds\]" $O/versions.log | head -9
BWTS_ROUND_TRACE=1 timeout -k 10 400 python bench.py --workload text --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --inverse-steps 1 --breakdown-steps 0 > $O/bench_text.log 2>&1; echo "rc=$?"; grep "rounds\]" $O/bench_text.log | head -10
exit 0
