# GPU session r02f: wide path (n > 2^32): forced small vs oracle, 1 GiB vs main path, 12 GiB properties; text re-measure
set -o pipefail
O=gpurun_out/r02f; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "wide or lf_walk" > $O/pytest_wide.log 2>&1; echo "pytest wide rc=$?"; tail -30 $O/pytest_wide.log
timeout -k 10 300 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $O/bench_text.json 2> $O/bench_text.err; echo "text rc=$?"
timeout -k 10 300 python tools/check_realtext.py > $O/realtext.log 2>&1; echo "realtext rc=$?"; tail -3 $O/realtext.log
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "versions rc=$?"; tail -3 $O/versions.log
