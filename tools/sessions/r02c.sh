# GPU session r02c: lean dense rounds + rank build v2, host path variants, inverse walker sweep
set -o pipefail
O=gpurun_out/r02c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
R=$GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_text -- python3 $R/bench.py --workload text --no-e2e --no-cpu-baseline --steps 2 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $R/$O/bench_text.json 2> $R/$O/bench_text.err; echo "prof rc=$?")
for w in 262144 524288 1048576 2097152; do BWTS_WALKERS=$w timeout -k 10 120 python tools/time_inverse.py zipf 30 7 8 2>&1 | sed "s/^/walkers $w: /"; done > $O/inverse_walkers.log; cat $O/inverse_walkers.log
timeout -k 10 200 python tools/time_host_path.py 30 > $O/host_path.log 2>&1; echo "host rc=$?"; cat $O/host_path.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
