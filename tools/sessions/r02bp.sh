# GPU session r02bp: moments by residue classes -- text / real text inverse (trace), tests, fuzz (main and 64-bit), 12 GiB
O=gpurun_out/r02bp; mkdir -p $O
BWTS_INV_TRACE=1 timeout -k 10 300 python bench.py --workload text --steps 1 --warmup 0 --inverse-steps 3 --breakdown-steps 0 --no-cpu-baseline --no-e2e > $O/bench_text.log 2>&1; grep "\[inverse\]" $O/bench_text.log | head -3; python3 -c "
import json
d=json.loads(open('$O/bench_text.log').read().strip().splitlines()[-1]); print('text inverse ms', d['inverse_ms_per_step'], d['roundtrip_exact'])"
BWTS_INV_TRACE=1 timeout -k 10 300 python tools/check_realtext.py 26 > $O/realtext.log 2>&1; grep "\[inverse\]\|roundtrip" $O/realtext.log | head -4
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverse or small or mid_size or tiny or long_cycle or low_entropy or reference_unbwts or golden or kat or wide" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 300 python tools/stress_random.py 300 12000 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=12 BWTS_WIDE_BUCKET=4096 timeout -k 10 300 python tools/stress_random.py 120 9000 > $O/stress_wide.log 2>&1; echo "wide fuzz rc=$?"; tail -1 $O/stress_wide.log
timeout -k 10 400 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; grep "inverse again\|round trip" $O/wide12.log | cut -c1-200
exit 0
