# GPU session r03bo: rocprofv3 kernel tables of the n > 2^32 runs (dna 12 GiB, text 6 GiB)
O=gpurun_out/r03bo; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/dna -o st -- python3 $R/tools/run_wide.py 12 dna > $R/$O/dna.log 2>&1; echo "dna rc=$?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/text -o st -- python3 $R/tools/run_wide.py 6 text > $R/$O/text.log 2>&1; echo "text rc=$?"
cd $R && ls $O/dna $O/text | head
