# GPU session r02bs: the numbers and profiles kept under profiles/ at the end of round 2
set -o pipefail
O=gpurun_out/r02bs; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --workload uniform256 --log2n 28 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-text > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
timeout -k 10 200 python bench.py --workload text --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/bench_text.json 2> $O/bench_text.err; echo "text rc=$?"
timeout -k 10 300 python tools/run_config4.py > $O/config4.json 2> $O/config4.err; echo "c4 rc=$?"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-e2e --no-text --steps 3 --warmup 1 --inverse-steps 2 --breakdown-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o st -- $B > $R/$O/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_text -o st -- python3 $R/bench.py --workload text --no-cpu-baseline --no-e2e --steps 3 --warmup 1 --inverse-steps 1 --breakdown-steps 0 > $R/$O/stats_text.log 2>&1; echo "stats text rc=$?"
cd $R && find $O -name "*.csv" | head -20
timeout -k 10 200 python tools/time_realtext.py 26 5 > $O/realtext.log 2>&1; echo "realtext rc=$?"
timeout -k 10 200 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "versions rc=$?"
exit 0
