# GPU session r03t: CLI cold-path breakdown (host-side costs now reported under BWTS_TIMINGS=1)
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 400 python tools/time_cli.py 30 2 > $O/cli.txt 2>&1; cat $O/cli.txt
