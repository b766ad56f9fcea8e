#!/usr/bin/env python3
"""Headline benchmark: BWTS build MB/s (+ inverse MB/s) on synthetic input, bit-exact round trip.

    python bench.py --gpus N --steps K --warmup W [--workload zipf|uniform256|dna] [--log2n L]

A "step" is one forward transform of one input that is already resident in HBM.  N > 1 runs
N independent replicas (one input per GPU, seed 1+rank; RCCL is only the barrier), launched
by torch.distributed.run.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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
    return {
        "value": round(n / 1e6 / fwd_s, 3), "unit": "MB/s", "cores": 1, "kind": "port",
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
                    "active_after_round0": 0, "unvisited": 0, "total_ms": 0.0}

    class _B:
        def free(self):
            pass

    def __init__(self, rank, ms):
        self.delay = (rank + 1) * ms / 1e3

    def alloc(self, n):
        return self._B()

    def generate(self, *a):
        pass

    def forward_device(self, *a):
        time.sleep(self.delay)

    inverse_device = forward_device

    def device_equal(self, *a):
        return True

    def timings(self):
        return self._T()

    def close(self):
        pass


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="zipf", choices=["zipf", "uniform256", "dna"])
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--inverse-steps", type=int, default=2)
    ap.add_argument("--cpu-sample-log2n", type=int, default=26)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--selftest-sleep-ms", type=float, default=0.0,
                    help="harness self-test (tests/test_dist_cpu.py): no GPU, no transform; every step sleeps "
                         "(rank+1) x this many ms so the rank/barrier/max-reduce/aggregate logic runs under gloo")
    args = ap.parse_args(argv)

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    selftest = args.selftest_sleep_ms > 0
    dev = "cpu" if selftest else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if selftest:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using %d" % (args.gpus, world, n_gpus), file=sys.stderr)

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
    ctx.generate(args.workload, 1 + rank, n, d_in)

    for _ in range(args.warmup):
        ctx.forward_device(d_in, n, d_out)

    # ---- timed region: exactly K forward steps ------------------------------------------
    agg = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.forward_device(d_in, n, d_out)          # synchronous: returns when d_out is complete
        tm = ctx.timings()                          # HIP-event times recorded on the engine's stream
        for name, st in tm.as_dict()["kernels"].items():
            a = agg.setdefault(name, {"ms": 0.0, "launches": 0, "alg_bytes": 0, "elems": 0})
            for k in a:
                a[k] += st[k]
    barrier()
    fwd_s = time.perf_counter() - t0
    fwd_info = tm.as_dict()

    # ---- inverse (reported beside the headline) + bit-exact round trip -----------------------
    ctx.inverse_device(d_out, n, d_back)            # warm
    roundtrip = ctx.device_equal(d_in, d_back, n)
    inv_agg = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.inverse_steps):
        ctx.inverse_device(d_out, n, d_back)
        ti = ctx.timings()
        for name, st in ti.as_dict()["kernels"].items():
            a = inv_agg.setdefault(name, {"ms": 0.0, "launches": 0, "alg_bytes": 0, "elems": 0})
            for k in a:
                a[k] += st[k]
    barrier()
    inv_s = time.perf_counter() - t0
    inv_info = ti.as_dict()

    times = torch.tensor([fwd_s, inv_s, 0.0 if roundtrip else 1.0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    fwd_s, inv_s, bad = [float(v) for v in times.tolist()]

    if rank == 0:
        ms_per_step = 1e3 * fwd_s / args.steps
        value = n_gpus * n / 1e6 / (fwd_s / args.steps)
        inv_value = n_gpus * n / 1e6 / (inv_s / max(args.inverse_steps, 1))

        def per_kernel(a):
            out = {}
            for name, st in a.items():
                if st["launches"] and st["ms"] > 0:
                    out[name] = {"ms_per_launch": round(st["ms"] / st["launches"], 4), "launches": st["launches"],
                                 "alg_GBps": round(st["alg_bytes"] / 1e9 / (st["ms"] / 1e3), 1)}
            return out

        # dominant kernel: the n-sized LSD passes of round 0 (one template variant, timed under its own class)
        sc = agg.get("radix_scatter_main") or agg.get("radix_scatter", {"ms": 0.0, "launches": 0, "alg_bytes": 0})
        achieved = sc["alg_bytes"] / 1e9 / (sc["ms"] / 1e3) if sc["ms"] > 0 else 0.0
        # which variant that was follows from its algorithmic bytes per element (20/18: packed streams, 26: wide pairs)
        per_elem = sc["alg_bytes"] // max(sc["launches"], 1) // max(n, 1) if sc["launches"] else 0
        kernel_label = {
            20: "radix_scatter_packed_kernel<false,false,true> (8-bit LSD pass over packed streams: key-low 4 B + value 4 B + key-high|carried byte 2 B)",
            18: "radix_scatter_packed_kernel<false,false,false> (8-bit LSD pass over packed streams: key 4 B + value 4 B + carried byte 1 B)",
        }.get(per_elem, "radix_scatter2_kernel<512,16,4,true,false> (8-bit LSD pass, key 8 B + value 4 B + carried byte)")
        traffic = None
        try:    # PMC bytes per launch, collected with rocprofv3 --pmc on this workload (profiles/), only valid for 2^30
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_radix_scatter.json")))
            if args.log2n == 30 and args.workload == "zipf" and pm.get("kernel", "").split("<")[0] == kernel_label.split("<")[0] \
                    and pm.get("alg_bytes_per_element") == per_elem:
                traffic = pm["hbm_bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "BWTS build MB/s on 1 GiB input (+ inverse MB/s); bit-exact round-trip",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s(n=2^%d, seed=1+rank) resident in HBM, forward BWTS per step" % (args.workload, args.log2n),
                       "bytes_per_gpu": n, "parallelism": "replicas x%d (one input per GPU, RCCL barrier only)" % n_gpus},
            "inverse_MBps": round(inv_value, 2), "inverse_ms_per_step": round(1e3 * inv_s / max(args.inverse_steps, 1), 3),
            "roundtrip_exact": bad == 0.0,
            "roofline": {"bound": "hbm", "kernel": kernel_label,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "alg_bytes_per_launch": sc["alg_bytes"] // max(sc["launches"], 1),
                         "ms_per_launch": round(sc["ms"] / max(sc["launches"], 1), 4), "launches": sc["launches"]},
            "forward": {"factors": fwd_info["factors"], "rounds": fwd_info["rounds"], "lyndon_rounds": fwd_info["lyndon_rounds"],
                        "key_symbols": fwd_info["key_symbols"], "key_bits": fwd_info["key_bits"],
                        "active_after_round0": fwd_info["active_after_round0"], "device_ms": round(fwd_info["total_ms"], 3),
                        "kernels": per_kernel(agg)},
            "inverse": {"cycles": inv_info["factors"], "unvisited": inv_info["unvisited"], "device_ms": round(inv_info["total_ms"], 3),
                        "kernels": per_kernel(inv_agg)},
        }
        if selftest:
            line["data"] = "selftest (no transform executed)"
        if n_gpus == 1 and not args.no_cpu_baseline and not selftest:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_log2n)
        print(json.dumps(line))
    for b in (d_in, d_out, d_back):
        b.free()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
