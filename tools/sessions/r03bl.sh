# GPU session r03bl: in/out in pinned host memory through the _device entry points (1 GiB zipf on the main path, 8 GiB dna on the wide path)
O=gpurun_out/r03bl; mkdir -p $O
timeout -k 10 300 python tools/check_host_resident.py 1 zipf > $O/host_resident_1GiB.txt 2>&1; rc=$?; echo "1 GiB rc=$rc"; tail -5 $O/host_resident_1GiB.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python tools/check_host_resident.py 8 dna > $O/host_resident_8GiB.txt 2>&1; echo "8 GiB rc=$?"; tail -5 $O/host_resident_8GiB.txt
