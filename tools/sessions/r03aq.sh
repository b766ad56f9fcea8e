# GPU session r03aq: the test selections of test_alternate_paths with every block poisoned (BWTS_POISON=1): does any path read memory nothing wrote?
O=gpurun_out/r03aq; mkdir -p $O
SEL="(small or mid_size or deep_repeats or dense_ties or dense_rounds or text_16MiB or reference_unbwts_vectors_through_cabi) and not alternate"
run() { # name, env assignments...
  name=$1; shift
  env BWTS_TEST_CHILD=1 BWTS_TEST_KNOBS=1 BWTS_POISON=1 BWTS_TRACE_ERRORS=1 "$@" timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SEL" > $O/$name.log 2>&1
  rc=$?; echo "$name rc=$rc $(tail -1 $O/$name.log)"; return $rc
}
run default &&
run tiles BWTS_DENSE=tiles &&
run tiles_order0 BWTS_DENSE=tiles BWTS_DENSE_ORDER=0 &&
run tiles_step2 BWTS_DENSE=tiles BWTS_DENSE_STEP=2 &&
run tiles_runs BWTS_DENSE=tiles BWTS_DENSE_RUNS=1 &&
run step2 BWTS_DENSE_STEP=2 &&
run legacy BWTS_DENSE=legacy &&
run legacy_noseg BWTS_DENSE=legacy BWTS_SEGSORT=0 &&
run varlen24 BWTS_VARLEN=1 BWTS_KEY_BITS=24 &&
run fixed2 BWTS_VARLEN=0 BWTS_KEY_SYMBOLS=2 &&
run lyndon_general BWTS_LYNDON=general &&
run emit_gather BWTS_EMIT=gather &&
run rx_pack0 BWTS_RX_PACK=0 &&
run rx_small0 BWTS_RX_SMALL=0 &&
run inv_log BWTS_INV_MARK=log &&
run inv_sentinel BWTS_INV_MARK=sentinel &&
run bytemark BWTS_BYTEMARK=1 &&
run split0 BWTS_SPLIT_LOG2=0
echo "session rc=$?"
