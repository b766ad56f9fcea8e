"""Condenses two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) of bench.py into the JSON files kept under
profiles/: per-kernel means, and the HBM bytes of the roofline kernels (bench.py reads those records, _pmc_traffic()).

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_dir> <round_tag> <workload> <log2n> <forwards> [<inverses>]

<forwards> / <inverses> = forward / inverse transforms the profiled command ran (warm-up included): kernels that run several
differently sized launches per transform (the rounds over the tied list) are reported per transform.
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are
reported in KB, and FETCH_SIZE counts half of the coalesced read bytes on gfx950, hence x2.
"""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_fingerprint as KF
CORRECTION = "gfx950: FETCH_SIZE reports 1/2 of coalesced read bytes (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE as is; units KB"
# (record name, kernel-name prefixes summed together, source file, algorithmic bytes per element, per what)
ROOFLINE = [
    ("radix_scatter", ["radix_scatter_packed_kernel<false, false, true, false>"], "radix.hip", 20, "launch"),
    ("walk", ["walk_record_kernel"], "inverse.hip", 6, "launch"),
    ("text_round", ["chunk_round_kernel", "chunk_apply_records_kernel"], "chunk_rounds.h", 32, "forward"),
]


def load(path):
    acc = collections.defaultdict(list)          # kernel -> [(grid, value)]
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    return acc


def demangled_map(fps):
    """{"name<template arguments>" as rocprofv3's demangled kernel names begin (behind an optional "void "): mangled symbol} for the
    library's kernels (kernel_fingerprint.pretty: the image has no c++filt)."""
    out = {}
    for sym in fps:
        name, targs = KF.pretty(sym)
        out[name + targs] = sym
    return out


def main():
    fetch, write, out_dir, tag, workload, log2n, forwards = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    inverses = int(sys.argv[8]) if len(sys.argv) > 8 else forwards
    allk = {}
    for name, acc in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for k, rows in acc.items():
            gmax = max(g for g, _ in rows)
            big = [v for g, v in rows if g == gmax]          # the largest launches of this kernel
            allk.setdefault(k[:120], {})[name] = {"launches_total": len(rows), "largest_launches": len(big), "mean_KB_largest": sum(big) / len(big),
                                                  "sum_KB": sum(v for _, v in rows)}
    json.dump(allk, open(os.path.join(out_dir, "%s_pmc_fetch_write_all_kernels_%s_2p%d.json" % (tag, workload, log2n)), "w"), indent=1)
    fps = KF.kernel_fingerprints()
    dem = demangled_map(fps)
    for rec, prefixes, source, per_elem, per in ROOFLINE:
        keys = [k for k in allk if any(k.startswith("void " + p) or k.startswith(p) for p in prefixes)]
        if not keys:
            continue
        # the record vouches for exactly the kernels whose launches it sums: their compiled code, by fingerprint
        symbols = {}
        for k in keys:
            kk = k[5:] if k.startswith("void ") else k
            hit = [m for d, m in dem.items() if kk.startswith(d + "(")]
            if len(hit) != 1:
                print("warning: %s: %d library kernels match %r" % (rec, len(hit), k), file=sys.stderr)
            for m in hit:
                symbols[m] = fps[m]
        out = {"kernel": prefixes[0] if len(prefixes) == 1 else " + ".join(prefixes), "alg_bytes_per_element": per_elem,
               "workload": "%s(2^%d, seed 1)" % (workload, log2n), "log2n": log2n, "source_file": "csrc/" + source,
               "kernel_symbols": symbols, "code_sha256": KF.combined(fps, symbols),
               "keyed_on": "sha256 over each kernel's machine code + kernel descriptor in libbwts_hip.so (tools/kernel_fingerprint.py)",
               "correction": CORRECTION, "collected": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes"}
        if per == "launch":
            f = sum(allk[k]["FETCH_SIZE"]["mean_KB_largest"] for k in keys)
            w = sum(allk[k]["WRITE_SIZE"]["mean_KB_largest"] for k in keys)
            out.update({"FETCH_SIZE_KB_reported": f, "WRITE_SIZE_KB_reported": w, "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)})
        else:
            div = forwards if rec != "walk" else inverses
            f = sum(allk[k]["FETCH_SIZE"]["sum_KB"] for k in keys) / div
            w = sum(allk[k]["WRITE_SIZE"]["sum_KB"] for k in keys) / div
            out.update({"FETCH_SIZE_KB_reported_per_forward": f, "WRITE_SIZE_KB_reported_per_forward": w, "transforms_profiled": div,
                        "hbm_bytes_per_forward": int(2 * f * 1024 + w * 1024)})
        json.dump(out, open(os.path.join(out_dir, "%s_pmc_traffic_%s_%s_2p%d.json" % (tag, rec, workload, log2n)), "w"), indent=1)
        print(rec, {k: v for k, v in out.items() if "KB" in k or "bytes" in k})


if __name__ == "__main__":
    main()
