# GPU session r02aa: erratum probe with the padded variant; wide paths after the workaround (regression test, fuzz in one context)
O=gpurun_out/r02aa; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/shift64_probe tools/probes/shift64_probe.hip 2> $O/build.log || { tail $O/build.log; exit 1; }
timeout -k 10 120 $O/shift64_probe > $O/probe.txt 2>&1; echo "probe rc=$?"; cat $O/probe.txt; rm -f $O/shift64_probe
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_path_small" > $O/wide_tests.log 2>&1; echo "wide tests rc=$?"; tail -4 $O/wide_tests.log
BWTS_FORCE_WIDE=1 BWTS_WIDE_SEG_LOG2=12 BWTS_WIDE_BUCKET=4096 timeout -k 10 400 python tools/stress_random.py 100 9000 > $O/stress_wide.log 2>&1; echo "wide fuzz rc=$?"; tail -4 $O/stress_wide.log
exit 0
