# GPU session r02e: key width for long repeats, LDS-window rank apply, 2-pass reorder; real-text checks
set -o pipefail
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_text -- python3 $R/bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $R/$O/bench_text.json 2> $R/$O/bench_text.err; echo "prof rc=$?")
timeout -k 10 300 python tools/check_realtext.py > $O/realtext.log 2>&1; echo "realtext rc=$?"; tail -5 $O/realtext.log
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "versions rc=$?"; tail -5 $O/versions.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
