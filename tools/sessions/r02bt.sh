# GPU session r02bt: splitter spacing after the walk changes (zipf 2^30, one box)
O=gpurun_out/r02bt; mkdir -p $O
timeout -k 10 300 python tools/time_inverse.py zipf 30 7 6 8 5 7 > $O/g.log 2>&1; echo "rc=$?"; cat $O/g.log
exit 0
