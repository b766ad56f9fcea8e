#!/bin/bash
# usage: tools/gpu_call.sh <timeout_s> <session script>   -- runs the script on a GPU box; waits while the pool has no free box (exit 3)
T=$1; S=$2
# what travels is what is built: never send sources newer than the library
make -s -C "$(dirname "$0")/../bijective-bwt_amd" -j8 all >/dev/null || { echo 'build failed'; exit 1; }
# ... and never a library whose kernels show the shift64 erratum pattern: a kernel edit can move a shift amount into the last VGPR
python3 "$(dirname "$0")/check_shift64.py" >/dev/null || { echo 'check_shift64 failed: run tools/check_shift64.py'; exit 1; }
make -s -C "$(dirname "$0")/../oracle" all >/dev/null || { echo 'oracle build failed'; exit 1; }
for try in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "bash $S"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
