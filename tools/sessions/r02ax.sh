# GPU session r02ax: host path by copy-worker count (fresh vs reused output buffers)
O=gpurun_out/r02ax; mkdir -p $O
nproc > $O/host.log; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))" >> $O/host.log
timeout -k 10 500 python tools/time_host_path.py 30 > $O/host_path.log 2>&1; echo "rc=$?"; cat $O/host.log; cat $O/host_path.log
exit 0
