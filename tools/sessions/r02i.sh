# GPU session r02i: wide inverse (n > 2^32): forced small, 1 GiB, 12 GiB round trip
set -o pipefail
O=gpurun_out/r02i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "wide or lf_walk" > $O/pytest_wide.log 2>&1; echo "pytest wide rc=$?"; tail -30 $O/pytest_wide.log
