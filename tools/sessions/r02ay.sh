# GPU session r02ay: host path, copy-worker counts in another order (6 first is where the pinned-block rows run)
O=gpurun_out/r02ay; mkdir -p $O
BWTS_SWEEP_THREADS=8,6,4,8,6,4 timeout -k 10 500 python tools/time_host_path.py 30 > $O/host_path.log 2>&1; echo "rc=$?"; grep "fresh out" $O/host_path.log
exit 0
