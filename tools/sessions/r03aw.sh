# GPU session r03aw: text bench line, kernel stats and PMC passes again (WIDE chunks)
O=gpurun_out/r03aw; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --workload text --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_text.json 2> $O/bench_text.err; echo "text rc=$?"
timeout -k 10 200 python tools/time_realtext.py > $O/realtext.txt 2>&1; echo "realtext rc=$?"
timeout -k 10 300 python tools/check_text_2p32.py 31 > $O/text_2p31.txt 2>&1; timeout -k 10 400 python tools/check_text_2p32.py 32 > $O/text_2p32.txt 2>&1; echo "big text rc=$?"
cd /tmp && export TMPDIR=/tmp
BT="python3 $R/bench.py --workload text --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --inverse-steps 1 --breakdown-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_text -o st -- $BT > $R/$O/stats_text.log 2>&1; echo "stats text rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_tf -o pf -- $BT > $R/$O/pmc_tf.log 2>&1; echo "pmc text f rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_tw -o pw -- $BT > $R/$O/pmc_tw.log 2>&1; echo "pmc text w rc=$?"
