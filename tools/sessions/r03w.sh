# GPU session r03w: batch entry points: parity test, then the bench with the batch leg
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or host_path or sink or cli" > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-text --no-cpu-baseline --steps 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03w/bench.json").read().strip().splitlines()[-1])
e=d["e2e"]
print({k:v for k,v in e.items() if "batch" in k or "host_forward" in k or "host_inverse" in k})
print(d.get("roofline_inverse"))
PY
