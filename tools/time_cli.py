"""GPU box: wall time and BWTS_TIMINGS breakdown of the two CLI programs on a zipf file in tmpfs (one-shot processes: HIP start-up,
allocation and code-object load are paid every time).    python tools/time_cli.py [log2n] [runs]"""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1 << log2n
pkg = ge.load_package()
ctx = pkg.Context(0)
d = ctx.alloc(n); ctx.generate("zipf", 1, n, d); x = d.download(); d.free(); ctx.close()
td = tempfile.mkdtemp(prefix="bwts_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    src, out, back = os.path.join(td, "in.bin"), os.path.join(td, "out.bwts"), os.path.join(td, "back.bin")
    x.tofile(src)
    env = dict(os.environ, BWTS_TIMINGS="1")
    pk = os.path.join(ROOT, "bijective-bwt_amd")
    for r in range(runs):
        for prog, a, b in (("mk_bwts", src, out), ("unbwts", out, back)):
            env["BWTS_T0_NS"] = str(time.clock_gettime_ns(time.CLOCK_MONOTONIC))      # the CLI reports what passed before its main()
            t0 = time.perf_counter()
            p = subprocess.run([os.path.join(pk, prog), a, b], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.perf_counter() - t0
            print("run %d %s: rc %d wall %.3f s = %.0f MB/s" % (r, prog, p.returncode, wall, n / 1e6 / wall))
            for l in p.stderr.decode().splitlines():
                if "time" in l: print("    " + l)
                if l.startswith("Process time") and "since launch" in l:
                    print("    after main() returned (exit handlers, device memory handed back by the driver): %.3f s" % (wall - float(l.rsplit("since launch", 1)[1].split()[0])))
            time.sleep(1.0)
    print("round trip exact:", bool(np.array_equal(np.fromfile(back, dtype=np.uint8), x)))
finally:
    shutil.rmtree(td, ignore_errors=True)
