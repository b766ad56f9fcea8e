# GPU session r03az: wide forward without the rank array (ties by direct comparison): parity runs, then dna 12 GiB forward
O=gpurun_out/r03az; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_path" > $O/pytest_wide.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_wide.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/run_wide.py 12 dna > $O/dna_12GiB.txt 2>&1; echo "dna12 rc=$?"; tail -8 $O/dna_12GiB.txt
