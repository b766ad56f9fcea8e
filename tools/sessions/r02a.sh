# GPU session r02a: full GPU suite, default bench, host-path sweep, rocprof of the text workload
set -o pipefail
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python tools/time_host_path.py 30 > $O/host_path.log 2>&1; echo "host rc=$?"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_text -- python3 $R/bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 0 --inverse-steps 1 > $R/$O/bench_text.json 2> $R/$O/bench_text.err; echo "prof rc=$?"
