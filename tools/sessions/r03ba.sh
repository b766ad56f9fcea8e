# GPU session r03ba: forward of dna(24 GiB) without the rank array, host-side checks (small rehearsal at 5 GiB first)
O=gpurun_out/r03ba; mkdir -p $O
timeout -k 10 400 python tools/check_wide_forward_only.py 5 dna 20000 > $O/fwd_only_5GiB.txt 2>&1; rc=$?; echo "5 GiB rc=$rc"; tail -6 $O/fwd_only_5GiB.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 1000 python tools/check_wide_forward_only.py 24 dna 100000 > $O/fwd_only_24GiB.txt 2>&1; echo "24 GiB rc=$?"; tail -6 $O/fwd_only_24GiB.txt
