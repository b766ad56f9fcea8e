# GPU session r03ab: text 2^31 with 48-bit (model) against 40-bit (packed) round-0 keys
O=gpurun_out/r03ab; mkdir -p $O
timeout -k 10 300 python tools/check_text_2p32.py 31 2>&1 | tail -3
BWTS_TEST_KNOBS=1 BWTS_KEY_BITS=40 timeout -k 10 300 python tools/check_text_2p32.py 31 2>&1 | tail -3
BWTS_TEST_KNOBS=1 BWTS_KEY_BITS=40 timeout -k 10 400 python tools/check_text_2p32.py 32 2>&1 | tail -3
