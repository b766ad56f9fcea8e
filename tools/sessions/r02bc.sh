# GPU session r02bc: latency of small and medium inputs
O=gpurun_out/r02bc; mkdir -p $O
timeout -k 10 300 python tools/time_small.py > $O/small.log 2>&1; echo "rc=$?"; cat $O/small.log
exit 0
