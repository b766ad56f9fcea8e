"""Manual check (GPU box): the transforms with input and output in PINNED HOST memory (bwts_host_alloc) handed to the _device entry points --
the kernels then read the text and write the result over PCIe, and only the working buffers live on the device.  This is the route for
inputs whose in/out copies do not fit beside the working set (n > 2^32 path: ~130 GiB of buffers whatever n is, when the ties are few).
Compared byte for byte with the device-resident run of the same input.      python tools/check_host_resident.py [gib] [kind]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = sys.argv[2] if len(sys.argv) > 2 else "dna"
n = gib << 30
ctx = pkg.Context(0)
a, b = ctx.alloc(n), ctx.alloc(n)
ctx.generate(kind, 1, n, a)
ctx.forward_device(a, n, b)
t0 = time.perf_counter(); ctx.forward_device(a, n, b); dev = time.perf_counter() - t0
t = ctx.timings()
print("device-resident forward: %.0f ms = %.2f GB/s (tied %d, %.0f GiB held by the context)" % (1e3 * dev, n / 1e9 / dev, t.active_after_round0, t.device_bytes / 2**30), flush=True)
want = b.download()
x = a.download()
a.free(); b.free()
t0 = time.perf_counter()
hin, pin = ctx.host_alloc(n)
hout, pout = ctx.host_alloc(n)
print("2 x %d GiB of pinned host memory: %.1f s" % (gib, time.perf_counter() - t0), flush=True)
hin[:] = x
t0 = time.perf_counter(); ctx.forward_device(pin, n, pout); host = time.perf_counter() - t0
ok = bool(np.array_equal(hout, want))
print("host-resident forward (text read and result written over PCIe): %.0f ms = %.2f GB/s, bytes equal: %s" % (1e3 * host, n / 1e9 / host, ok), flush=True)
# inverse: the same way (B in pinned memory, T out to pinned memory)
hout[:] = 0
hin[:] = want
t0 = time.perf_counter(); ctx.inverse_device(pin, n, pout); hinv = time.perf_counter() - t0
ok2 = bool(np.array_equal(hout, x))
print("host-resident inverse: %.0f ms = %.2f GB/s, text restored: %s" % (1e3 * hinv, n / 1e9 / hinv, ok2), flush=True)
sys.exit(0 if ok and ok2 else 1)
