"""Manual timing (GPU box): 1 GiB of real text (tests/realtext.py corpus_big), resident in HBM: wall time per forward / inverse, the
kernel-class spans of one more call, round trip; with `hash` also the per-root sha-256 of the corpus (which part of the image differs
from the build container's?).      python tools/time_realtext_big.py [reps] [hash]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import realtext
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if "hash" in sys.argv:
    for root in (realtext.BIG_ROOTS if False else ("/usr/lib/python3/dist-packages", "/usr/lib/python3.10", "/usr/share/doc", "/usr/include", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include", "/opt/rocm/share")):
        h, tot, cnt = hashlib.sha256(), 0, 0
        sub = {}
        for d, dirs, files in os.walk(root):
            dirs.sort()
            for f in sorted(files):
                if not f.endswith((".py", ".txt", ".h", ".hpp", ".md", ".rst", ".c", ".json", ".html")): continue
                p = os.path.join(d, f)
                if os.path.islink(p): continue
                try: b = open(p, "rb").read()
                except OSError: continue
                h.update(b); tot += len(b); cnt += 1
                top = os.path.relpath(p, root).split(os.sep)[0]
                s = sub.setdefault(top, [hashlib.sha256(), 0]); s[0].update(b); s[1] += len(b)
        print("root", root, cnt, "files", tot, "bytes", h.hexdigest()[:16], flush=True)
        if root.endswith("local/lib/python3.10/dist-packages") or root.endswith("rocm/share"):
            for top in sorted(sub): print("   ", top, sub[top][1], sub[top][0].hexdigest()[:12])
x = np.frombuffer(realtext.corpus_big(1 << 30), dtype=np.uint8)
n = len(x)
print("corpus", n, hashlib.sha256(x.tobytes()).hexdigest(), flush=True)
pkg = ge.load_package(); ctx = pkg.Context(0)
a, b, c = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
a.upload(x)
ctx.forward_device(a.ptr, n, b.ptr)
ts, ti = [], []
for r in range(reps):
    t0 = time.perf_counter(); ctx.forward_device(a.ptr, n, b.ptr); ts.append(time.perf_counter() - t0)
t = ctx.timings()
print("real text n=%d: forward wall ms %s  best %.2f = %.2f GB/s   rounds %d tied %d" % (n, [round(1e3 * v, 2) for v in ts], 1e3 * min(ts), n / 1e9 / min(ts), t.rounds, t.active_after_round0))
for r in range(reps):
    t0 = time.perf_counter(); ctx.inverse_device(b.ptr, n, c.ptr); ti.append(time.perf_counter() - t0)
print("inverse wall ms", [round(1e3 * v, 2) for v in ti], "round trip exact:", ctx.device_equal(a, c, n))
ctx.set_timing(2)
ctx.forward_device(a.ptr, n, b.ptr)
k = ctx.timings().as_dict()
print("level-2 spans: %.2f ms in kernels (%d launches)" % (sum(v["ms"] for v in k["kernels"].values()), sum(v["launches"] for v in k["kernels"].values())), {cn: round(v["ms"], 2) for cn, v in k["kernels"].items()})
print("round_active", k.get("round_active"))
