# GPU session r02k: larger-group path proportional to its elements; steady-state wide timings
set -o pipefail
O=gpurun_out/r02k; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "small or mid_size or deep_repeats or dense_ties or text or structured or alternate" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 3 --warmup 1 --breakdown-steps 1 --inverse-steps 1 > $O/bench_text.json 2> $O/bench_text.err; echo "text rc=$?"
BWTS_TIMINGS=0 timeout -k 10 300 python - > $O/realtext_untimed.log 2>&1 <<'PY'
import re, runpy, sys
sys.argv = ["check_realtext.py"]
src = open("tools/check_realtext.py").read().replace("ctx.set_timing(2)", "ctx.set_timing(0)")
exec(compile(src, "tools/check_realtext.py", "exec"))
src = open("tools/check_versions_text.py").read().replace("ctx.set_timing(2)", "ctx.set_timing(0)")
exec(compile(src, "tools/check_versions_text.py", "exec"))
PY
echo "realtext rc=$?"; grep -E "real text|versions" $O/realtext_untimed.log | cut -c1-200
timeout -k 10 300 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; grep again $O/wide12.log
