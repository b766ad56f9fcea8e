# GPU session r03x: batch pipeline stage times under different copy-out modes
O=gpurun_out/r03x; mkdir -p $O
for mode in kernel dma; do
  echo "== D2H $mode"; BWTS_D2H=$mode BWTS_BATCH_TRACE=1 timeout -k 10 300 python tools/time_batch.py 30 8 2>&1 | tee $O/batch_$mode.txt
done
echo "== D2H split 50"; BWTS_D2H_SPLIT=50 BWTS_BATCH_TRACE=1 timeout -k 10 300 python tools/time_batch.py 30 8 2>&1 | tee $O/batch_split.txt
echo "== copy threads 3"; BWTS_COPY_THREADS=3 BWTS_BATCH_TRACE=1 timeout -k 10 300 python tools/time_batch.py 30 8 2>&1 | tee $O/batch_t3.txt
