# GPU session r03l: phase shares inside chunk_round_kernel (build with -DCH_PROFILE), text 2^28
O=gpurun_out/r03l; mkdir -p $O
BWTS_ROUND_TRACE=1 timeout -k 10 300 python - > $O/trace.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests")
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
n = 1 << 28
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate("text", 1, n, a)
for rep in range(2):
    ctx.forward_device(a, n, b)
    print("rep", rep, "total ms", ctx.timings().total_ms, flush=True)
PY
tail -40 $O/trace.txt
