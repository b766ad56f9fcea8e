# GPU session r02bd: 64-bit forward path with buckets of 3 * 2^30 elements (12 GiB DNA)
O=gpurun_out/r02bd; mkdir -p $O
BWTS_WIDE_BUCKET=3221225472 timeout -k 10 400 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; cut -c1-300 $O/wide12.log
exit 0
