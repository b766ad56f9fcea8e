# GPU session r03bd: phase shares inside the chunk round kernel (library built with -DCH_PROFILE: clock64 marks), text 2^30 and real text
O=gpurun_out/r03bd; mkdir -p $O
BWTS_LIB_OVERRIDE=$PWD/tools/ab/libbwts_chprofile.so BWTS_ROUND_TRACE=1 timeout -k 10 300 python tools/time_realtext.py 26 1 2>&1 | grep -E "phase shares|round [0-9]+ h" | head -24 | tee $O/realtext_phases.txt
BWTS_LIB_OVERRIDE=$PWD/tools/ab/libbwts_chprofile.so BWTS_ROUND_TRACE=1 timeout -k 10 300 python bench.py --workload text --no-e2e --no-cpu-baseline --steps 1 --warmup 0 --inverse-steps 1 --breakdown-steps 0 2>&1 | grep -E "phase shares|round [0-9]+ h" | head -24 | tee $O/text_phases.txt
