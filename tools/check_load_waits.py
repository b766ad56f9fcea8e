"""Device-code reading aid (no GPU needed): compiles the library's .hip files to gfx950 assembly and lists, per kernel, how many of its
global loads are followed within two instructions by `s_waitcnt vmcnt(0)` -- the shape a load under a bounds test takes
(`x = i < n ? a[i] : 0` becomes a branch around the load with a wait for its data right behind it: ONE load in flight per wave,
however far the loop was unrolled; DESIGN.md section 10, profiles/history/r04_load_scheduling.md).  A hit is a place to LOOK, not a
verdict: a chain of dependent loads reads the same way.

    python tools/check_load_waits.py [min_hits] [file.hip ...]
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bijective-bwt_amd", "csrc")


def device_asm(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, path], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def scan(txt):
    res = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        loads = serial = 0
        for i, l in enumerate(lines):
            if l.startswith(("global_load", "buffer_load", "flat_load")):
                loads += 1
                if any(i + k < len(lines) and lines[i + k].startswith("s_waitcnt") and "vmcnt(0)" in lines[i + k] for k in (1, 2)):
                    serial += 1
        res.append((serial, loads, name))
    return res


def main(argv):
    min_hits = int(argv[0]) if argv and argv[0].isdigit() else 4
    files = [a for a in argv if a.endswith(".hip")] or [os.path.join(CSRC, f) for f in ("forward.hip", "radix.hip", "inverse.hip")]
    for f in files:
        for serial, loads, name in sorted(scan(device_asm(f)), reverse=True):
            if serial >= min_hits:
                print("%-12s %3d of %3d loads  %s" % (os.path.basename(f), serial, loads, name[:140]))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
