# GPU session r02ao: faster fused column scan -- sort tests, fuzz, real text timing
O=gpurun_out/r02ao; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "radix or small or mid_size or text_16MiB" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 300 python tools/stress_random.py 150 7000 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
timeout -k 10 300 python tools/time_realtext.py 26 5 > $O/realtext.log 2>&1; echo "rc=$?"; head -1 $O/realtext.log | cut -c1-200
timeout -k 10 300 python tools/check_versions_text.py > $O/versions.log 2>&1; echo "rc=$?"; tail -2 $O/versions.log | head -1 | cut -c1-110
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e > $O/bench.json 2> $O/bench.err; python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('zipf fwd', d['ms_per_step'], 'inv', d['inverse_ms_per_step'], 'text', d['text']['forward_ms'], 'traffic', d['roofline']['traffic'])"
exit 0
