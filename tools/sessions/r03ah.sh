# GPU session r03ah: full gpu suite on the final code of round 3, then the PMC passes of the default bench again (inverse.hip changed)
O=gpurun_out/r03ah; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?
tail -6 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-e2e --no-text --steps 3 --warmup 1 --inverse-steps 2 --breakdown-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -o pf -- $B > $R/$O/pmc_f.log 2>&1; echo "pmc f rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -o pw -- $B > $R/$O/pmc_w.log 2>&1; echo "pmc w rc=$?"
