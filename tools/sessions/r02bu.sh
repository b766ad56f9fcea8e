# GPU session r02bu: walker count after the walk changes
O=gpurun_out/r02bu; mkdir -p $O
timeout -k 10 300 python tools/diag/walkers.py > $O/w.log 2>&1; echo "rc=$?"; cat $O/w.log
exit 0
