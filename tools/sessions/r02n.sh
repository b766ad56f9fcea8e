# GPU session r02n: activation rounds (runs) in the dense path
set -o pipefail
O=gpurun_out/r02n; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "small or mid_size or deep_repeats or dense_ties or text or structured or alternate or threshold or golden or cabi or tiny" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 300 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 3 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $O/bench_text.json 2> $O/bench_text.err; echo "text rc=$?"
for t in check_realtext check_versions_text; do sed 's/ctx.set_timing(2)/ctx.set_timing(0)/' tools/$t.py > tools/_untimed_$t.py; timeout -k 10 300 python tools/_untimed_$t.py 2>&1 | grep -E "real text|versions|roundtrip|oracle" | cut -c1-160; rm -f tools/_untimed_$t.py; done
timeout -k 10 300 python tools/check_text_2p32.py 31 2>&1 | tail -3
