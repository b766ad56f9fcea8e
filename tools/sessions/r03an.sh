# GPU session r03an: wide forward with the tied list in blocks and the rounds in parts: forced-wide parity runs, then text(6 GiB)
O=gpurun_out/r03an; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_path" > $O/pytest_wide.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_wide.log
timeout -k 10 600 python tools/run_wide.py 6 text > $O/text_6GiB.txt 2>&1; echo "text rc=$?"; tail -12 $O/text_6GiB.txt
