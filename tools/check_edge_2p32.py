"""Manual check (GPU box): sizes just below 2^32 that are not tile-aligned -- device-side round trip, timings, counters."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
for n, kind, seed in (((1 << 32) - 12345, "dna", 3), ((1 << 32) - 1, "zipf", 3), (1 << 32, "dna", 3)):
    a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ctx.generate(kind, seed, n, a)
    for rep in range(2):
        t0 = time.time(); ctx.forward_device(a, n, b); tf = time.time() - t0
        t = ctx.timings().as_dict()
        print(kind, n, "rep", rep, "forward wall %.0f ms device %.0f ms" % (1e3 * tf, t["total_ms"]), {k: t[k] for k in ("factors", "rounds", "lyndon_rounds", "key_bits", "active_after_round0")},
              {k: round(v["ms"]) for k, v in t["kernels"].items()}, flush=True)
    t0 = time.time(); ctx.inverse_device(b, n, c); ti = time.time() - t0
    print("   inverse %.0f ms roundtrip" % (1e3 * ti), ctx.device_equal(a, c, n), flush=True)
    for x in (a, b, c): x.free()
