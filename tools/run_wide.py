"""Manual check (GPU box): the n > 2^32 paths on dna / text / zipf of some GiB: per-class device times of forward and inverse, round trip.
    python tools/run_wide.py [gib] [kind]      (default 12 dna)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 12
kind = sys.argv[2] if len(sys.argv) > 2 else "dna"
n = gib << 30
ctx = pkg.Context(0); ctx.set_timing(2)
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate(kind, 1, n, a)
t0 = time.perf_counter(); ctx.forward_device(a, n, b); dt = time.perf_counter() - t0
k = ctx.timings().as_dict()
print("forward %.0f ms wall, device %.0f ms" % (1e3 * dt, k["total_ms"]), {x: round(v["ms"], 1) for x, v in k["kernels"].items()},
      "rounds", k["rounds"], "tied", k["active_after_round0"], "per round", [int(v) for v in k.get("round_active", [])][:40], "factors", k["factors"], "device GiB %.0f" % (k["device_bytes"] / 2**30), flush=True)
c = ctx.alloc(n)
t0 = time.perf_counter(); ctx.inverse_device(b, n, c); dt = time.perf_counter() - t0
k = ctx.timings().as_dict()
print("inverse %.0f ms wall, device %.0f ms" % (1e3 * dt, k["total_ms"]), {x: round(v["ms"], 1) for x, v in k["kernels"].items()},
      "cycles", k["factors"], "unreached", k["unvisited"], flush=True)
print("round trip exact:", ctx.device_equal(a, c, n))
for name, fn, src, dst in (("forward", ctx.forward_device, a, b), ("inverse", ctx.inverse_device, b, c)):
    t0 = time.perf_counter(); fn(src, n, dst); dt = time.perf_counter() - t0
    print("%s again (arenas in place): %.0f ms wall = %.2f GB/s" % (name, 1e3 * dt, n / 1e9 / dt), flush=True)
