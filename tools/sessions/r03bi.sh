# GPU session r03bi: dna 12 GiB with the largest buckets (no rank array), wide parity runs
O=gpurun_out/r03bi; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_path" > $O/pytest_wide.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_wide.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/run_wide.py 12 dna > $O/dna_12GiB.txt 2>&1; echo "dna12 rc=$?"; tail -6 $O/dna_12GiB.txt
