# GPU session r02cf: PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of the final build -- all kernels, the walk among them
set -o pipefail
O=gpurun_out/r02cf; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-e2e --no-text --steps 3 --warmup 1 --inverse-steps 2 --breakdown-steps 0"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -o pf -- $B > $R/$O/pmc_f.log 2>&1; echo "pmc f rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -o pw -- $B > $R/$O/pmc_w.log 2>&1; echo "pmc w rc=$?"
exit 0
