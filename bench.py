#!/usr/bin/env python3
"""Headline benchmark: BWTS build MB/s (+ inverse MB/s) on synthetic input, bit-exact round trip.

    python bench.py --gpus N --steps K --warmup W [--workload zipf|uniform256|dna|text] [--log2n L]

A "step" is one forward transform of one input that is already resident in HBM.  N > 1 runs N independent
replicas (one input per GPU, seed 1+rank; RCCL is only the barrier and the max-over-ranks of the times): either
under an outer `python -m torch.distributed.run ... bench.py --gpus N`, or -- when WORLD_SIZE is not set --
bench.py starts those N ranks itself as a child process before anything here touches a GPU.
Rank 0 prints ONE JSON line.  At N = 1 the line also carries: the text workload beside the headline one, the
host-buffer path (bwts_forward / bwts_inverse on unpinned caller memory) and the CLI wall time (`e2e`), and the
CPU baseline.
"""
import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# A process that never touches the GPU: it starts the CLI programs for the e2e leg (the parent has an open GPU
# context by then) and times them.  Protocol: one JSON request per line on stdin, one JSON answer per line on stdout.
_HELPER = r'''
import json, subprocess, sys, time
for line in sys.stdin:
    req = json.loads(line)
    env = req.get("env")
    if env is not None:
        env["BWTS_T0_NS"] = str(time.clock_gettime_ns(time.CLOCK_MONOTONIC))      # the CLI reports what passed before its main()
    t0 = time.perf_counter()
    p = subprocess.run(req["cmd"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    wall = time.perf_counter() - t0
    print(json.dumps({"rc": p.returncode, "wall_s": wall, "stderr": p.stderr.decode(errors="replace")[-2000:]}), flush=True)
'''


def cpu_baseline(kind, sample_log2n):
    """Oracle (CPU restatement, 1 core) on a bounded sample of the same workload."""
    import numpy as np
    import oracle_lib as O
    n = 1 << sample_log2n
    x = O.generate(kind, n, 1)
    t0 = time.perf_counter()
    y, phases = O.forward_timed(x)
    fwd_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    back = O.inverse(y)
    inv_s = time.perf_counter() - t0
    assert np.array_equal(back, x)
    # the reference's own suffix sorter (mk_bwts_sa.c:48), timed on the same sample if the box happens to have the library
    # (this image does not: then the row says so)
    dss = None
    try:
        import ctypes
        import ctypes.util
        name = ctypes.util.find_library("divsufsort")
        if name:
            L = ctypes.CDLL(name)
            sa = np.empty(n, dtype=np.int32)
            t0 = time.perf_counter()
            rc = L.divsufsort(ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(sa.ctypes.data), ctypes.c_int32(n))
            dss = {"divsufsort_s": round(time.perf_counter() - t0, 3), "rc": int(rc)}
    except Exception as e:          # absent or unusable: reported, not fatal
        dss = {"error": repr(e)}
    return {
        "value": round(n / 1e6 / fwd_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
        "libdivsufsort": dss if dss is not None else "not installed on this box (the reference links it: Makefile:4)",
        "sample": "first 2^%d bytes of the %s(seed=1) stream; oracle/bwts_oracle.c forward (own SA-IS suffix sorter + "
                  "reference fix-up), single thread" % (sample_log2n, kind),
        "forward_s": round(fwd_s, 3),
        "phase_s": {"suffix_sort": round(phases[0], 3), "isa": round(phases[1], 3), "fix": round(phases[2], 3),
                    "generate": round(phases[3], 3)},
        "inverse_value": round(n / 1e6 / inv_s, 3), "inverse_s": round(inv_s, 3),
        "host_cpus": os.cpu_count(),
    }


class _SleepEngine:
    """Stand-in used ONLY by --selftest-sleep-ms: runs no transform and touches no GPU."""

    class _T:
        total_ms = 0.0

        def as_dict(self):
            return {"kernels": {}, "factors": 0, "rounds": 0, "lyndon_rounds": 0, "key_symbols": 0, "key_bits": 0,
                    "active_after_round0": 0, "unvisited": 0, "total_ms": 0.0, "round_active": []}

    class _B:
        def free(self):
            pass

    def __init__(self, rank, ms):
        self.delay = (rank + 1) * ms / 1e3

    def alloc(self, n):
        return self._B()

    def generate(self, *a):
        pass

    def set_timing(self, level):
        pass

    def forward_device(self, *a):
        time.sleep(self.delay)

    inverse_device = forward_device

    def forward_batch(self, arrays):
        time.sleep(self.delay * len(arrays))
        return [a.copy() for a in arrays]

    def device_equal(self, *a):
        return True

    def timings(self):
        return self._T()

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _accumulate(agg, tm):
    for name, st in tm.as_dict()["kernels"].items():
        a = agg.setdefault(name, {"ms": 0.0, "launches": 0, "alg_bytes": 0, "elems": 0})
        for k in a:
            a[k] += st[k]


def _per_kernel(a):
    out = {}
    for name, st in a.items():
        if st["launches"] and st["ms"] > 0:
            out[name] = {"ms_per_launch": round(st["ms"] / st["launches"], 4), "launches": st["launches"],
                         "alg_GBps": round(st["alg_bytes"] / 1e9 / (st["ms"] / 1e3), 1)}
    return out


def _real_text(ctx, d_in, d_out, d_back, cap, big=False):
    """Real text through the device path: best of 5 forwards and inverses (wall time per synchronous call), round trip.  big = False: the
    53.6 MiB corpus of the gpu suite; True: 1 GiB of the same kind of material from more of the image (tests/realtext.py corpus_big)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from realtext import corpus, corpus_big
    x = np.frombuffer(corpus_big(1 << 30) if big else corpus(1 << 26), dtype=np.uint8)
    m = int(x.size)
    if m < (1 << 20) or m > cap or (big and m != 1 << 30):
        return {"skipped": "corpus of %d bytes" % m}
    d_in.upload(x)
    ctx.set_timing(0)
    ctx.forward_device(d_in, m, d_out)
    tf, ti = [], []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.forward_device(d_in, m, d_out); tf.append(time.perf_counter() - t0)
    info = ctx.timings().as_dict()
    ctx.inverse_device(d_out, m, d_back)
    for _ in range(5):
        t0 = time.perf_counter(); ctx.inverse_device(d_out, m, d_back); ti.append(time.perf_counter() - t0)
    return {"workload": ("Python packages, ROCm headers and data files, source and documentation files of this image, concatenated (tests/realtext.py "
                         "corpus_big)" if big else "source and documentation files of this image, concatenated (tests/realtext.py)"), "bytes": m,
            "sha256_in": hashlib.sha256(x.tobytes()).hexdigest(),
            "forward_ms": round(1e3 * min(tf), 2), "forward_MBps": round(m / 1e6 / min(tf), 1),
            "inverse_ms": round(1e3 * min(ti), 2), "inverse_MBps": round(m / 1e6 / min(ti), 1),
            "roundtrip_exact": bool(ctx.device_equal(d_in, d_back, m)), "rounds": info["rounds"], "tied_after_round0": info["active_after_round0"]}


_FPS = None


def _fingerprints():
    global _FPS
    if _FPS is None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_fingerprint as KF
        _FPS = (KF, KF.kernel_fingerprints())
    return _FPS


def _pmc_traffic(kernel_prefix, workload, log2n, per_elem=None):
    """HBM bytes of a kernel from the committed rocprofv3 --pmc passes (profiles/*pmc_traffic*.json, made by
    tools/pmc_summary.py).  A record names the kernel symbols it summed and the sha256 of their COMPILED code (machine code + kernel
    descriptor, read out of libbwts_hip.so: tools/kernel_fingerprint.py): a kernel whose code has changed since reports null instead
    of a stale figure, and an edit elsewhere in the kernel's source file does not retire the record."""
    try:
        KF, fps = _fingerprints()
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if not (name.endswith(".json") and "pmc_traffic" in name):
                continue
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
            if not str(pm.get("workload", "zipf")).startswith(workload) or pm.get("log2n", 30) != log2n:
                continue
            if not pm.get("kernel", "").split("<")[0].startswith(kernel_prefix.split("<")[0]):
                continue
            if per_elem is not None and pm.get("alg_bytes_per_element") != per_elem:
                continue
            syms = pm.get("kernel_symbols")
            if not syms or any(s not in fps for s in syms):       # (records without fingerprints cannot vouch for today's kernel)
                continue
            if KF.combined(fps, syms) != pm.get("code_sha256"):
                continue
            return pm.get("hbm_bytes_per_launch", pm.get("hbm_bytes_per_forward")), name
    except Exception:
        pass
    return None, None


def _config5_leg(ctx, rank, n, items, workload, barrier, selftest):
    """BASELINE config 5's real data flow on this rank's GPU: `items` input files of n bytes in tmpfs, mmapped (the CLIs' map_in),
    through bwts_forward_batch (copies of the neighbouring items overlapped with the transform) into fresh host buffers.  Timed between
    two barriers; the caller takes the max over ranks.  Returns (seconds, info).  What limits it on a full node is what SURVEY 8(e)
    predicts: page-cache / mmap and PCIe traffic of all ranks at once, not the GPUs."""
    import numpy as np
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    td = tempfile.mkdtemp(prefix="bwts_c5_r%d_" % rank, dir=base)
    try:
        paths = []
        if selftest:
            for i in range(items):
                p = os.path.join(td, "in%d.bin" % i)
                np.zeros(min(n, 1 << 16), dtype=np.uint8).tofile(p)
                paths.append(p)
        else:
            tmp = ctx.alloc(n)
            for i in range(items):
                ctx.generate(workload, 1 + rank * items + i, n, tmp)
                p = os.path.join(td, "in%d.bin" % i)
                tmp.download().tofile(p)
                paths.append(p)
            tmp.free()
        maps = [np.memmap(p, dtype=np.uint8, mode="r") for p in paths]
        if not selftest:
            ctx.forward_batch([maps[0][: min(n, 1 << 24)]] * 2)        # warm: the batch's staging rings, workers, second pair of device buffers
        barrier()
        t0 = time.perf_counter()
        outs = ctx.forward_batch(maps)
        barrier()
        dt = time.perf_counter() - t0
        ok = True
        if not selftest:
            back = ctx.inverse(outs[0])                                 # one item checked per rank (outside the timed region)
            ok = bool(np.array_equal(back, maps[0]))
        info = {"items_per_gpu": items, "bytes_each": int(maps[0].size), "files": "tmpfs (%s), mmapped read-only" % (base or td),
                "outputs": "fresh (untouched) host buffers allocated inside the call", "first_item_roundtrip_exact": ok}
        del maps, outs
        return dt, info
    finally:
        shutil.rmtree(td, ignore_errors=True)


LINE_FILL_CEILING = 54e9       # random 128-byte line fills per second this chip sustains (tools/micro/random_read.hip)


def _roofline(kernel, st, traffic, traffic_src, note, per_forward=None):
    """roofline object of one kernel class from its summed HIP-event time and algorithmic bytes (timed region)."""
    ms = st.get("ms", 0.0)
    launches = st.get("launches", 0)
    achieved = st.get("alg_bytes", 0) / 1e9 / (ms / 1e3) if ms > 0 else 0.0
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
         "launches": launches, "timed": note}
    if per_forward:
        r["alg_bytes_per_forward"] = st.get("alg_bytes", 0) // per_forward
        r["ms_per_forward"] = round(ms / per_forward, 3)
    else:
        r["alg_bytes_per_launch"] = st.get("alg_bytes", 0) // max(launches, 1)
        r["ms_per_launch"] = round(ms / max(launches, 1), 4)
    return r


def _cli_runs(helper, src, out, rt, n, pause_s=0.0):
    """mk_bwts src -> out, [pause,] unbwts out -> rt through the helper process: walls, rates, the programs' own phase lines."""
    env = dict(os.environ, BWTS_TIMINGS="1")

    def ask(cmd):
        helper.stdin.write(json.dumps({"cmd": cmd, "env": env}) + "\n")
        helper.stdin.flush()
        return json.loads(helper.stdout.readline())

    pkg_dir = os.path.join(ROOT, "bijective-bwt_amd")
    a = ask([os.path.join(pkg_dir, "mk_bwts"), src, out])
    if pause_s: time.sleep(pause_s)
    b = ask([os.path.join(pkg_dir, "unbwts"), out, rt])

    def phases(r):
        lines = [l for l in r["stderr"].splitlines() if " time" in l][-8:]
        for l in lines:
            if l.startswith("Process time") and "since launch" in l:        # what follows main(): exit handlers, the driver taking the memory back
                lines.append("After main() %.3f" % (r["wall_s"] - float(l.rsplit("since launch", 1)[1].split()[0])))
                break
        return lines
    return {"ok": a["rc"] == 0 and b["rc"] == 0,
            "wall_s": {"mk_bwts": round(a["wall_s"], 3), "unbwts": round(b["wall_s"], 3)},
            "MBps": (round(n / 1e6 / a["wall_s"], 1) if a["rc"] == 0 else None, round(n / 1e6 / b["wall_s"], 1) if b["rc"] == 0 else None),
            "phases": {"mk_bwts": phases(a), "unbwts": phases(b)}, "stderr": (a["stderr"] + b["stderr"])[-500:]}


def _cli_first(ctx, helper, d_in, n, workload, seed):
    """The two programs on a tmpfs file BEFORE this process has freed any device memory, each on a device at rest (a pause before
    unbwts): what a user who runs `mk_bwts file` meets.  Then the pair once more back to back: device memory that has been freed -- by
    a process that has just exited, or by a living one -- is scrubbed by the driver before it is used again, 30-40 ms per GiB, and
    whoever allocates next waits for it: unbwts started right behind mk_bwts waits for mk_bwts' 27 GiB.
    The files stay until _e2e has compared them with the device path."""
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    td = tempfile.mkdtemp(prefix="bwts_bench_", dir=base)
    import atexit
    atexit.register(shutil.rmtree, td, True)         # (tmpfs is memory: the files go whatever becomes of the run)
    src, out, rt = os.path.join(td, "in.bin"), os.path.join(td, "out.bwts"), os.path.join(td, "back.bin")
    ctx.generate(workload, seed, n, d_in)
    d_in.download().tofile(src)
    r = _cli_runs(helper, src, out, rt, n, pause_s=2.0)
    time.sleep(2.0)
    out2, rt2 = out + ".2", rt + ".2"
    r2 = _cli_runs(helper, src, out2, rt2, n)
    for f in (out2, rt2):
        if os.path.exists(f): os.remove(f)
    return {"runs": r, "back_to_back": r2, "dir": td, "src": src, "out": out, "rt": rt, "base": base or td}


def _e2e(ctx, helper, d_in, d_out, n, workload, cli_pre=None):
    """The path the CLIs take: caller-owned, unpinned host buffers (bwts_forward / bwts_inverse), and the mk_bwts / unbwts
    programs on files in tmpfs, started by the helper process."""
    import numpy as np
    res = {}
    x = d_in.download()
    ymem = d_out.download()                     # forward output of the timed steps (device path)

    def timed(fn, src, reps=3):
        best, tot = None, 0.0
        for _ in range(reps):
            out = np.empty(n, dtype=np.uint8)   # fresh, never-touched output: the first-touch faults are the caller's reality
            t0 = time.perf_counter()
            fn(src, out)
            dt = time.perf_counter() - t0
            tot += dt
            best = dt if best is None or dt < best else best
        return out, tot / reps, best

    ctx.forward_into(x, np.empty(n, dtype=np.uint8))        # warm: staging ring, copy workers, device in/out buffers
    y, mean_s, best_s = timed(ctx.forward_into, x)
    tm = ctx.timings()
    res["host_forward_MBps"] = round(n / 1e6 / mean_s, 1)
    res["host_forward_best_MBps"] = round(n / 1e6 / best_s, 1)
    res["host_forward_ms"] = {"wall": round(1e3 * mean_s, 2), "h2d": round(tm.h2d_ms, 2), "device": round(tm.total_ms, 2),
                              "d2h": round(tm.d2h_ms, 2)}
    res["host_forward_equals_device_path"] = bool(np.array_equal(y, ymem))
    back, mean_s, best_s = timed(ctx.inverse_into, y)
    tm = ctx.timings()
    res["host_inverse_MBps"] = round(n / 1e6 / mean_s, 1)
    res["host_inverse_ms"] = {"wall": round(1e3 * mean_s, 2), "h2d": round(tm.h2d_ms, 2), "device": round(tm.total_ms, 2),
                              "d2h": round(tm.d2h_ms, 2)}
    res["host_roundtrip_exact"] = bool(np.array_equal(back, x))
    res["host_buffers"] = "numpy arrays (malloc, unpinned); output allocated fresh for every call"
    del back

    # a batch of independent inputs on this one GPU (BASELINE config 5's data flow per GPU): bwts_forward_batch overlaps item k+1's
    # copy in and item k-1's copy out with item k's transform
    if n <= (1 << 30):
        items = 8
        xs = [x]
        tmp = ctx.alloc(n)
        for sd in range(2, items + 1):
            ctx.generate(workload, sd, n, tmp)
            xs.append(tmp.download())
        tmp.free()
        ctx.forward_batch(xs[:2])                     # warm: the batch's own staging rings, workers and second pair of device buffers
        t0 = time.perf_counter()
        ys = ctx.forward_batch(xs)
        bf = time.perf_counter() - t0
        t0 = time.perf_counter()
        backs = ctx.inverse_batch(ys)
        bi = time.perf_counter() - t0
        res["batch_host_forward_MBps"] = round(items * n / 1e6 / bf, 1)
        res["batch_host_inverse_MBps"] = round(items * n / 1e6 / bi, 1)
        res["batch"] = {"items": items, "bytes_each": n, "forward_wall_ms": round(1e3 * bf, 1), "inverse_wall_ms": round(1e3 * bi, 1),
                        "first_item_equals_device_path": bool(np.array_equal(ys[0], ymem)),
                        "roundtrip_exact": bool(all(np.array_equal(b, xi) for b, xi in zip(backs, xs))),
                        "inputs": "%s seeds 1..%d, unpinned numpy arrays; outputs allocated (untouched) inside the call" % (workload, items)}
        del xs, ys, backs

    if helper is not None and cli_pre is not None:
        try:
            r = cli_pre["runs"]
            res["cli_wall_MBps"], res["cli_inverse_wall_MBps"] = r["MBps"]
            res["cli_wall_s"] = r["wall_s"]
            res["cli_phases"] = r["phases"]
            if r["ok"]:
                res["cli_output_equals_device_path"] = bool(np.array_equal(np.fromfile(cli_pre["out"], dtype=np.uint8), ymem))
                res["cli_roundtrip_exact"] = bool(np.array_equal(np.fromfile(cli_pre["rt"], dtype=np.uint8), x))
            else:
                res["cli_error"] = r["stderr"]
            res["cli_files"] = ("tmpfs (%s); wall time of the whole process: HIP start-up, context, mmap, transform, write; run before this "
                                "process had freed any device memory, 2 s apart (back to back: cli_wall_back_to_back_s)" % cli_pre["base"])
            r2 = cli_pre["back_to_back"]
            res["cli_wall_back_to_back_s"] = r2["wall_s"]
            res["cli_phases_back_to_back"] = {k: [l for l in v if l.startswith(("Process time", "Start-up time", "After main"))] for k, v in r2["phases"].items()}
        finally:
            shutil.rmtree(cli_pre["dir"], ignore_errors=True)
    return res


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="zipf", choices=["zipf", "uniform256", "dna", "text"])
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--inverse-steps", type=int, default=2)
    ap.add_argument("--breakdown-steps", type=int, default=2, help="extra, untimed-for-the-headline steps with every kernel class timed")
    ap.add_argument("--cpu-sample-log2n", type=int, default=26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-buffer / CLI leg")
    ap.add_argument("--no-text", action="store_true", help="skip the text workload reported beside the headline one")
    ap.add_argument("--config5-items", type=int, default=4, help="files per GPU in the host-path leg of an N > 1 run (BASELINE config 5); 0 = skip")
    ap.add_argument("--config5-leg", action="store_true", help="run that leg at N = 1 too")
    ap.add_argument("--selftest-sleep-ms", type=float, default=0.0,
                    help="harness self-test (tests/test_dist_cpu.py): no GPU, no transform; every step sleeps "
                         "(rank+1) x this many ms so the rank/barrier/max-reduce/aggregate logic runs under gloo")
    args = ap.parse_args(argv)
    raw_args = list(sys.argv[1:] if argv is None else argv)

    # ---- N > 1 without an outer launcher: start the N ranks as a child, before anything here touches a GPU ----------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + raw_args
        env = dict(os.environ)
        env.setdefault("OMP_NUM_THREADS", "1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        rc = subprocess.call(cmd, env=env)          # the ranks inherit stdout: rank 0's JSON line is this process's line
        sys.exit(rc)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    selftest = args.selftest_sleep_ms > 0
    n_gpus = world

    helper = None
    if n_gpus == 1 and not selftest and not args.no_e2e:
        helper = subprocess.Popen([sys.executable, "-c", _HELPER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    import torch
    dev = "cpu" if selftest else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if selftest:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            t = torch.zeros(1, device=dev)
            dist.all_reduce(t)          # RCCL used as a barrier only
        if not selftest:
            torch.cuda.synchronize()

    n = 1 << args.log2n
    if selftest:
        ctx = _SleepEngine(rank, args.selftest_sleep_ms)
    else:
        import __graft_entry__ as ge
        ctx = ge.load_package().Context(local_rank)
    d_in = ctx.alloc(n)
    d_out = ctx.alloc(n)
    d_back = ctx.alloc(n)
    cli_pre = None
    if helper is not None:
        try:
            cli_pre = _cli_first(ctx, helper, d_in, n, args.workload, 1 + rank)
        except Exception as e:
            print("bench.py: CLI leg failed: %r" % (e,), file=sys.stderr)

    def measure(workload, steps, warmup, inverse_steps, breakdown_steps):
        """K timed forward steps (HIP events only on the dominant kernel), the inverse + round trip, then the breakdown."""
        ctx.generate(workload, 1 + rank, n, d_in)
        ctx.set_timing(1)
        for _ in range(warmup):
            ctx.forward_device(d_in, n, d_out)
        main_agg = {}
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.forward_device(d_in, n, d_out)          # synchronous: returns when d_out is complete
            tm = ctx.timings()                          # HIP-event time of the dominant kernel, on the engine's stream
            _accumulate(main_agg, tm)
        barrier()
        fwd_s = time.perf_counter() - t0
        fwd_info = tm.as_dict()

        ctx.inverse_device(d_out, n, d_back)            # warm
        roundtrip = ctx.device_equal(d_in, d_back, n)
        inv_main = {}
        barrier()
        t0 = time.perf_counter()
        for _ in range(inverse_steps):
            ctx.inverse_device(d_out, n, d_back)
            ti = ctx.timings()
            _accumulate(inv_main, ti)
        barrier()
        inv_s = time.perf_counter() - t0
        inv_info = ti.as_dict()

        # every kernel class timed: its own pass, outside the two timed regions
        agg, inv_agg = {}, {}
        ctx.set_timing(2)
        for _ in range(breakdown_steps):
            ctx.forward_device(d_in, n, d_out)
            _accumulate(agg, ctx.timings())
            ctx.inverse_device(d_out, n, d_back)
            _accumulate(inv_agg, ctx.timings())
        ctx.set_timing(0)
        return {"fwd_s": fwd_s, "inv_s": inv_s, "roundtrip": roundtrip, "fwd_info": fwd_info, "inv_info": inv_info,
                "main": main_agg, "inv_main": inv_main, "agg": agg, "inv_agg": inv_agg}

    m = measure(args.workload, args.steps, args.warmup, args.inverse_steps, args.breakdown_steps)
    # N > 1: the leg that measures config 5's real data flow (files in tmpfs -> bwts_forward_batch -> host buffers), every rank at once
    c5_s, c5_info = 0.0, None
    if (n_gpus > 1 or args.config5_leg) and args.config5_items > 0:
        c5_s, c5_info = _config5_leg(ctx, rank, n, args.config5_items, args.workload, barrier, selftest)
        if not c5_info["first_item_roundtrip_exact"]:
            m["roundtrip"] = False
    times = torch.tensor([m["fwd_s"], m["inv_s"], 0.0 if m["roundtrip"] else 1.0, c5_s], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    fwd_s, inv_s, bad, c5_s = [float(v) for v in times.tolist()]

    rc = 0
    if rank == 0:
        ms_per_step = 1e3 * fwd_s / args.steps
        value = n_gpus * n / 1e6 / (fwd_s / args.steps)
        inv_value = n_gpus * n / 1e6 / (inv_s / max(args.inverse_steps, 1))
        fwd_info, inv_info = m["fwd_info"], m["inv_info"]

        # dominant kernel: the n-sized LSD passes of round 0 (one template variant, timed under its own class in the timed region)
        sc = m["main"].get("radix_scatter_main") or {"ms": 0.0, "launches": 0, "alg_bytes": 0}
        achieved = sc["alg_bytes"] / 1e9 / (sc["ms"] / 1e3) if sc["ms"] > 0 else 0.0
        # which variant that was follows from its algorithmic bytes per element (20/18: packed streams, 26: wide pairs)
        per_elem = sc["alg_bytes"] // max(sc["launches"], 1) // max(n, 1) if sc["launches"] else 0
        kernel_label = {
            20: "radix_scatter_packed_kernel<false,false,true> (8-bit LSD pass over packed streams: key-low 4 B + value 4 B + key-high|carried byte 2 B)",
            18: "radix_scatter_packed_kernel<false,false,false> (8-bit LSD pass over packed streams: key 4 B + value 4 B + carried byte 1 B)",
        }.get(per_elem, "radix_scatter2_kernel<512,16,4,true,false> (8-bit LSD pass, key 8 B + value 4 B + carried byte)")
        traffic, traffic_src = _pmc_traffic(kernel_label, args.workload, args.log2n, per_elem)
        size_label = "%d GiB" % (n >> 30) if n >= 1 << 30 and n % (1 << 30) == 0 else "%d MiB" % (n >> 20) if n >= 1 << 20 else "%d B" % n
        line = {
            "metric": "BWTS build MB/s on %s input (+ inverse MB/s); bit-exact round-trip" % size_label,
            "value": round(value, 2) if bad == 0.0 else None, "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s(n=2^%d, seed=1+rank) resident in HBM, forward BWTS per step" % (args.workload, args.log2n),
                       "bytes_per_gpu": n, "parallelism": "replicas x%d (one input per GPU, RCCL barrier only)" % n_gpus},
            "inverse_MBps": round(inv_value, 2), "inverse_ms_per_step": round(1e3 * inv_s / max(args.inverse_steps, 1), 3),
            "roundtrip_exact": bad == 0.0,
            "roofline": {"bound": "hbm", "kernel": kernel_label,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": sc["alg_bytes"] // max(sc["launches"], 1),
                         "ms_per_launch": round(sc["ms"] / max(sc["launches"], 1), 4), "launches": sc["launches"],
                         "timed": "HIP events on the engine's stream inside the timed region (this class only; the per-class "
                                  "breakdown under forward.kernels comes from %d separate steps)" % args.breakdown_steps},
            "forward": {"factors": fwd_info["factors"], "rounds": fwd_info["rounds"], "lyndon_rounds": fwd_info["lyndon_rounds"],
                        "key_symbols": fwd_info["key_symbols"], "key_bits": fwd_info["key_bits"],
                        "active_after_round0": fwd_info["active_after_round0"], "round_active": fwd_info.get("round_active"),
                        "device_ms": round(fwd_info["total_ms"], 3), "kernels": _per_kernel(m["agg"])},
            "inverse": {"cycles": inv_info["factors"], "unvisited": inv_info["unvisited"], "device_ms": round(inv_info["total_ms"], 3),
                        "walk_ms_timed_region": round(m["inv_main"].get("walk", {}).get("ms", 0.0) / max(args.inverse_steps, 1), 3),
                        "kernels": _per_kernel(m["inv_agg"])},
        }
        # second dominant kernel: the inverse's splitter walk -- bound by line fills, not bytes (one 128-byte fill per 4-byte LF read)
        wk = m["inv_main"].get("walk")
        if wk and wk.get("ms", 0) > 0:
            wt, wsrc = _pmc_traffic("walk_record_kernel", args.workload, args.log2n)
            rw = _roofline("walk_record_kernel<3,16> (one LF chase per splitter, symbols recorded in 64-byte blocks; 6 n algorithmic bytes: "
                           "4 LF + 1 symbol + 1 output)", wk, wt, wsrc,
                           "HIP events on the engine's stream inside the timed inverse region (%d calls)" % args.inverse_steps)
            if wt:
                fills = wt / 128.0 / (rw["ms_per_launch"] / 1e3)
                rw["line_fills_per_s"] = round(fills / 1e9, 1)
                rw["line_fill_ceiling_per_s"] = LINE_FILL_CEILING / 1e9
                rw["line_fill_frac"] = round(fills / LINE_FILL_CEILING, 3)
                rw["traffic_over_algorithmic"] = round(wt / max(rw["alg_bytes_per_launch"], 1), 1)
            line["roofline_inverse"] = rw
        if c5_info is not None:
            # whole job: every rank's files over the slowest rank's wall time (value stays the HBM-resident rate the contract names)
            c5_bytes = n_gpus * c5_info["items_per_gpu"] * c5_info["bytes_each"]
            line["config5_host_path"] = dict(c5_info, wall_ms=round(1e3 * c5_s, 2), aggregate_MBps=round(c5_bytes / 1e6 / c5_s, 3),
                                             what="per rank: files -> mmap -> bwts_forward_batch -> host buffers, all ranks at once, barrier to barrier, max over ranks")
        if selftest:
            line["data"] = "selftest (no transform executed)"
        if n_gpus == 1 and not selftest:
            if not args.no_e2e:
                try:
                    line["e2e"] = _e2e(ctx, helper, d_in, d_out, n, args.workload, cli_pre)
                except Exception as e:       # the headline stands on its own; a failure here is reported, not hidden
                    line["e2e"] = {"error": repr(e)}
            if not args.no_text and args.workload != "text":
                # the repeat-rich text workload beside the headline one, same size
                t = measure("text", 2, 1, 1, 1)
                ti = t["fwd_info"]
                line["text"] = {
                    "workload": "text(n=2^%d, seed=1): zipf stream with back-references of 16 B .. 64 KiB" % args.log2n,
                    "forward_MBps": round(n / 1e6 / (t["fwd_s"] / 2), 1), "forward_ms": round(1e3 * t["fwd_s"] / 2, 2),
                    "inverse_MBps": round(n / 1e6 / t["inv_s"], 1), "inverse_ms": round(1e3 * t["inv_s"], 2),
                    "roundtrip_exact": bool(t["roundtrip"]), "rounds": ti["rounds"], "key_bits": ti["key_bits"], "factors": ti["factors"],
                    "round_active": ti.get("round_active"), "kernels": _per_kernel(t["agg"]),
                }
                rk = t["main"].get("round")
                if rk and rk.get("ms", 0) > 0:
                    # the text regime's dominant kernel: the group-local round over the chunked tied list (all launches of a forward)
                    rt, rsrc = _pmc_traffic("chunk_round_kernel", "text", args.log2n)
                    line["text"]["roofline"] = _roofline(
                        "chunk_round_kernel<true,3,true> + chunk_apply_moves_kernel (per list element and round: position + head in, three "
                        "successor ranks gathered, position + head out, rank updated = 32 algorithmic bytes)", rk, rt, rsrc,
                        "HIP events on the engine's stream inside the timed region of the text leg (2 forwards)", per_forward=2)
                if not t["roundtrip"]:
                    bad = 1.0
                # real text beside the synthetic one (the reference's own workload is enwik8, Makefile:35-38, which no box has): source and
                # documentation files of this image, as in the gpu suite (tests/realtext.py builds it; no oracle involved here)
                try:
                    line["text"]["real"] = _real_text(ctx, d_in, d_out, d_back, n)
                    if line["text"]["real"].get("roundtrip_exact") is False:
                        bad = 1.0
                except Exception as e:
                    line["text"]["real"] = {"error": repr(e)}
                # ... and 1 GiB of real text, the size the metric is quoted on (no enwik9 on any box; this is what the image itself holds)
                if n >= (1 << 30):
                    try:
                        line["text"]["real_1GiB"] = _real_text(ctx, d_in, d_out, d_back, n, big=True)
                        if line["text"]["real_1GiB"].get("roundtrip_exact") is False:
                            bad = 1.0
                    except Exception as e:
                        line["text"]["real_1GiB"] = {"error": repr(e)}
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_log2n)
        print(json.dumps(line), flush=True)
        if bad != 0.0:
            print("bench.py: round trip NOT exact", file=sys.stderr)
            rc = 1
    for b in (d_in, d_out, d_back):
        b.free()
    ctx.close()
    if helper is not None:
        helper.stdin.close()
        helper.wait(timeout=30)
    if dist is not None:
        dist.destroy_process_group()
    if bad != 0.0:
        rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
