# GPU session r03as: kernel table of the real-text forward (rocprofv3 --kernel-trace --stats), with the round trace
O=gpurun_out/r03as; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BWTS_ROUND_TRACE=1 timeout -k 10 300 python tools/time_realtext.py 26 3 > $O/realtext_trace.txt 2>&1; echo "rc=$?"; tail -30 $O/realtext_trace.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o rt -- python tools/time_realtext.py 26 3 > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
find $O/prof -name "*kernel_stats*" | head -3
