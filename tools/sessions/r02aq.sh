# GPU session r02aq: where the real-text forward's time goes (wall vs kernel spans, kernel table)
O=gpurun_out/r02aq; mkdir -p $O
timeout -k 10 300 python tools/time_realtext.py 26 5 > $O/realtext.log 2>&1; echo "rc=$?"; cat $O/realtext.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/time_realtext.py 26 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo $f; head -32 $f | cut -c1-150,400-520
exit 0
