# GPU session r02j: full suite at HEAD; kernel tables for the real-text checks; wide-path class times
set -o pipefail
O=gpurun_out/r02j; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout -k 10 300 python tools/run_wide.py 12 > $O/wide12.log 2>&1; echo "wide rc=$?"; cat $O/wide12.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_realtext -o st -- python3 $R/tools/check_realtext.py > $R/$O/realtext.log 2>&1; echo "realtext rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_versions -o st -- python3 $R/tools/check_versions_text.py > $R/$O/versions.log 2>&1; echo "versions rc=$?"
