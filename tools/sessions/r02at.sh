# GPU session r02at: text 2^31 / 2^32 through the main path with the quadrupled step; 12 GiB DNA through the 64-bit paths (timings)
O=gpurun_out/r02at; mkdir -p $O
timeout -k 10 300 python tools/check_text_2p32.py 31 > $O/text_2p31.log 2>&1; echo "text31 rc=$?"; cat $O/text_2p31.log
timeout -k 10 400 python tools/check_text_2p32.py 32 > $O/text_2p32.log 2>&1; echo "text32 rc=$?"; cat $O/text_2p32.log
timeout -k 10 400 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; cut -c1-260 $O/wide12.log
exit 0
