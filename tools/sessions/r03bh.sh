# GPU session r03bh: bench.py default run with the real-text leg
O=gpurun_out/r03bh; mkdir -p $O
s=$(date +%s); timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03bh/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["text"]["forward_ms"], d["text"].get("real"))
PY
