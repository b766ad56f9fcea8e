# GPU session r03by: final sanity of HEAD on the GPU: smoke(), a parity subset, the default bench line
O=gpurun_out/r03by; mkdir -p $O
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "small or known or structured or cli_end_to_end or batch_equals or pinned_host or chunk_rounds or text_16MiB or real_text" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03by/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roundtrip_exact"], d["text"]["forward_ms"], d["text"]["real"]["forward_ms"], d["inverse_ms_per_step"])
PY
