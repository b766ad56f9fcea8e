# GPU session r03ad: kernel stats of the text workload (where do the order kernels stand)
O=gpurun_out/r03ad; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_text -o st -- python3 $R/bench.py --workload text --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --inverse-steps 1 --breakdown-steps 0 > $R/$O/stats_text.log 2>&1; echo "stats rc=$?"
cd $R
python - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/r03ad/stats_text/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:28]:
    print("%-70s calls %4s total %8.2f ms avg %8.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
