# GPU session r03p: derived PMC metrics of chunk_round_kernel on text(2^28): what bounds it?
O=gpurun_out/r03p; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for M in VALUBusy MemUnitBusy MemUnitStalled LDSBankConflict VALUUtilization; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $R/$O/$M -o p -- python3 $R/tools/pmc_text_probe.py > $R/$O/$M.log 2>&1; echo "$M rc=$?"
done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/$O/sq1 -o p -- python3 $R/tools/pmc_text_probe.py > $R/$O/sq1.log 2>&1; echo "sq1 rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/$O/sq2 -o p -- python3 $R/tools/pmc_text_probe.py > $R/$O/sq2.log 2>&1; echo "sq2 rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY --output-format csv -d $R/$O/sq3 -o p -- python3 $R/tools/pmc_text_probe.py > $R/$O/sq3.log 2>&1; echo "sq3 rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r03p/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "chunk_round_kernel" in k or "radix_scatter2_kernel<512, 16, 4, true, false, false" in k:
            a=acc[k[:40]][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,d in acc.items():
        print(f.split("/")[2], k, {c:(n, round(v/n,2)) for c,(n,v) in d.items()})
PY
