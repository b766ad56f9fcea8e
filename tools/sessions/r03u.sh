# GPU session r03u: default bench line (headline + e2e + text + cpu baseline) on the current build
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03u/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "inv", d["inverse_ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["ms_per_launch"])
print("e2e", json.dumps(d["e2e"], indent=1))
print("text", d["text"]["forward_ms"], d["text"]["rounds"], d["text"]["key_bits"])
print("cpu", d["cpu_baseline"]["value"])
PY
