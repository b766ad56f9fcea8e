# GPU session r02bw: moment classes by n (4096 above 2^30) -- tests, fuzz, config 4, zipf 2^30 inverse
O=gpurun_out/r02bw; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverse or small or mid_size or tiny or long_cycle or low_entropy or reference_unbwts or golden or boundary or 2p32 or config4" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 300 python tools/stress_random.py 200 13000 > $O/stress.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/stress.log
BWTS_INV_TRACE=1 timeout -k 10 300 python tools/run_config4.py > $O/config4.json 2> $O/config4.err; grep "\[inverse\]" $O/config4.err | tail -2; python3 -c "
import json
c=json.loads(open('$O/config4.json').read().strip().splitlines()[-1]); print('c4 fwd', c['forward_ms'], 'inv', c['inverse_ms'], c['roundtrip_exact'], 'unvisited', c['unvisited'])"
timeout -k 10 300 python tools/time_inverse.py zipf 30 > $O/inv30.log 2>&1; cat $O/inv30.log
exit 0
