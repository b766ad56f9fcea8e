# GPU session r02m: wide path with 2^31 buckets; text at 2^31 / 2^32 second-call timings
set -o pipefail
O=gpurun_out/r02m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "wide or lf_walk" > $O/pytest_wide.log 2>&1; echo "pytest wide rc=$?"; tail -8 $O/pytest_wide.log
timeout -k 10 300 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; cat $O/wide12.log
timeout -k 10 300 python tools/check_text_2p32.py 31 > $O/text_2p31.log 2>&1; echo "text31 rc=$?"; cat $O/text_2p31.log
timeout -k 10 300 python tools/check_text_2p32.py 32 > $O/text_2p32.log 2>&1; echo "text32 rc=$?"; cat $O/text_2p32.log
