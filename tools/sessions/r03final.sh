# GPU session r03final (run again as r03final2 after the WIDE chunks and the wide forward without ranks): the numbers and profiles kept under profiles/ (round 3)
O=gpurun_out/r03final2; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --workload uniform256 --log2n 28 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-text > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
timeout -k 10 200 python bench.py --workload text --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_text.json 2> $O/bench_text.err; echo "text rc=$?"
timeout -k 10 300 python tools/run_config4.py > $O/config4.json 2> $O/config4.err; echo "c4 rc=$?"
timeout -k 10 200 python tools/time_realtext.py > $O/realtext.txt 2>&1; echo "realtext rc=$?"
timeout -k 10 300 python tools/time_cli.py 30 2 > $O/cli.txt 2>&1; echo "cli rc=$?"
BWTS_BATCH_TRACE=1 timeout -k 10 300 python tools/time_batch.py 30 8 > $O/batch.txt 2>&1; echo "batch rc=$?"
timeout -k 10 300 python tools/check_text_2p32.py 31 > $O/text_2p31.txt 2>&1; timeout -k 10 400 python tools/check_text_2p32.py 32 > $O/text_2p32.txt 2>&1; echo "big text rc=$?"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-e2e --no-text --steps 3 --warmup 1 --inverse-steps 2 --breakdown-steps 0"
BT="python3 $R/bench.py --workload text --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --inverse-steps 1 --breakdown-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o st -- $B > $R/$O/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_text -o st -- $BT > $R/$O/stats_text.log 2>&1; echo "stats text rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -o pf -- $B > $R/$O/pmc_f.log 2>&1; echo "pmc f rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -o pw -- $B > $R/$O/pmc_w.log 2>&1; echo "pmc w rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_tf -o pf -- $BT > $R/$O/pmc_tf.log 2>&1; echo "pmc text f rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_tw -o pw -- $BT > $R/$O/pmc_tw.log 2>&1; echo "pmc text w rc=$?"
cd $R && find $O -name "*.csv" | head -20
