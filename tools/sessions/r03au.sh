# GPU session r03au: round trace of real text and text 2^28 with WIDE chunks
O=gpurun_out/r03au; mkdir -p $O
BWTS_ROUND_TRACE=1 timeout -k 10 300 python tools/time_realtext.py 26 1 2>&1 | grep "chunks\]" | head -14
