# GPU session r03ar: stress of the wide rounds in parts
O=gpurun_out/r03ar; mkdir -p $O
STRESS_BUDGET_S=420 timeout -k 10 600 python tools/stress_wide_parts.py 400 0 > $O/stress_wide_parts.txt 2>&1; echo "rc=$?"; tail -8 $O/stress_wide_parts.txt; grep -c OK $O/stress_wide_parts.txt; grep -c refused $O/stress_wide_parts.txt
