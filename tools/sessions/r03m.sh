# GPU session r03m: chunk rounds with early gathers: parity subset, text bench, kernel stats and PMC traffic on the text workload
O=gpurun_out/r03m; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or text_16MiB or deep_repeats or mid_size" > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --workload text --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > $O/bench_text.json 2> $O/bench_text.err || { tail -5 $O/bench_text.err; exit 1; }
timeout -k 10 200 python tools/time_realtext.py > $O/realtext.txt 2>&1; cat $O/realtext.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload text --no-cpu-baseline --no-e2e --steps 2 --warmup 1 --inverse-steps 1 --breakdown-steps 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_text -o st -- $B > $R/$O/stats_text.log 2>&1; echo "stats rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_f -o pf -- $B > $R/$O/pmc_f.log 2>&1; echo "pmc f rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_w -o pw -- $B > $R/$O/pmc_w.log 2>&1; echo "pmc w rc=$?"
cd $R
python - <<'PY'
import json, csv, glob, collections
d=json.loads(open("gpurun_out/r03m/bench_text.json").read().strip().splitlines()[-1])
print("text", d["ms_per_step"], d["roundtrip_exact"], d["forward"]["rounds"], {k:(v["ms_per_launch"],v["launches"]) for k,v in d["forward"]["kernels"].items()})
for tag in ("pmc_f/**/*counter_collection.csv", "pmc_w/**/*counter_collection.csv"):
    for f in glob.glob("gpurun_out/r03m/"+tag, recursive=True):
        acc=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            a=acc[r["Kernel_Name"][:60]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
        print(f)
        for k,(c,v) in sorted(acc.items(), key=lambda kv:-kv[1][1])[:12]: print("  %-60s calls %5d  sum KB %.0f" % (k,c,v))
PY
find $O -name "*stats*.csv" | head
