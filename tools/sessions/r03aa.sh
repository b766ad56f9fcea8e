# GPU session r03aa: text at 2^31 and 2^32 through the chunk rounds (u32 edge: positions up to 2^32 - 1)
O=gpurun_out/r03aa; mkdir -p $O
BWTS_ROUND_TRACE=1 timeout -k 10 300 python tools/check_text_2p32.py 31 > $O/text_2p31.txt 2>&1; grep -v "^\[chunks\] round" $O/text_2p31.txt | tail -6
BWTS_ROUND_TRACE=1 BWTS_TRACE_ERRORS=1 timeout -k 10 400 python tools/check_text_2p32.py 32 > $O/text_2p32.txt 2>&1; grep -v "^\[chunks\] round" $O/text_2p32.txt | tail -8
BWTS_ROUND_TRACE=1 timeout -k 10 200 python tools/time_realtext.py 26 2 2>&1 | grep -v "^\[chunks\] phase" | head -30 > $O/realtext_trace.txt; head -14 $O/realtext_trace.txt
