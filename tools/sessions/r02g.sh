# GPU session r02g: full GPU suite incl. wide path; text workloads
set -o pipefail
O=gpurun_out/r02g; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
timeout -k 10 300 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $O/bench_text.json 2> $O/bench_text.err; echo "text rc=$?"
timeout -k 10 300 python tools/check_realtext.py > $O/realtext.log 2>&1; echo "realtext rc=$?"; tail -3 $O/realtext.log
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "versions rc=$?"; tail -3 $O/versions.log
