# GPU session r03v: full gpu suite on the current build (chunk rounds, lean arena, lazy CLI output), then the default bench
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03v/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "inv", d["inverse_ms_per_step"], "roofline", d["roofline"]["frac"])
e=d["e2e"]; print("e2e host fwd", e["host_forward_MBps"], "cli", e["cli_wall_s"]); print("\n".join(e["cli_phases"]))
print("text", d["text"]["forward_ms"], d["text"]["rounds"], d["text"]["key_bits"])
PY
