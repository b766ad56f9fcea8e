"""Build-time check of the device code: 64-bit shifts whose shift amount sits in the wave's last allocated VGPR.

Found in round 2 on MI355X (tools/README.md): v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 read a wrong shift
amount when the amount register is the last VGPR of the wave's allocation (register number = 7 mod 8 and nothing allocated
behind it) and other waves share the SIMD.  LLVM knows this as the gfx90a "shift64 high register" erratum and moves the
amount to another register there; for gfx950 it does not.  This script compiles every .hip file of the library to device
assembly and reports such instructions; `make check-isa` runs it and the test suite asserts a clean report.

usage: python tools/check_shift64.py [file.hip ...]        exit code 1 when an instruction is found
"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bijective-bwt_amd", "csrc")
SHIFT = re.compile(r"^\s*(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\w*\s+v\[\d+:\d+\],\s*v(\d+)\s*,")
LABEL = re.compile(r"^([A-Za-z_][\w.$]*):")
NEXT = re.compile(r"\.amdhsa_next_free_vgpr\s+(\d+)")
ACC = re.compile(r"\.amdhsa_accum_offset\s+(\d+)")
KERNEL = re.compile(r"\.amdhsa_kernel\s+(\S+)")


def hipcc():
    """the compiler the Makefile uses ($HIPCC, else /opt/rocm/bin/hipcc); None when there is none"""
    cand = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    return cand if (os.path.isfile(cand) and os.access(cand, os.X_OK)) or shutil.which(cand) else None


def device_asm(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [hipcc() or "hipcc", "--offload-arch=" + os.environ.get("ARCH", "gfx950"), "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC, "--cuda-device-only", "-S", "-o", out, path]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("%s failed (%d):\n%s" % (" ".join(cmd), r.returncode, r.stdout.decode(errors="replace")[-4000:]))
        return open(out).read().splitlines()


def scan(lines):
    """-> list of (kernel, instruction text, amount register, arch VGPRs)"""
    shifts, cur, found = {}, None, []
    for ln in lines:
        m = LABEL.match(ln)
        if m and not m.group(1).startswith(".L"):
            cur = m.group(1)
            shifts.setdefault(cur, [])
        m = SHIFT.match(ln)
        if m and cur:
            shifts[cur].append((int(m.group(2)), ln.strip()))
    kern, nxt = None, {}
    for ln in lines:
        m = KERNEL.search(ln)
        if m: kern = m.group(1)
        m = NEXT.search(ln)
        if m and kern: nxt.setdefault(kern, {})["next"] = int(m.group(1))
        m = ACC.search(ln)
        if m and kern: nxt.setdefault(kern, {})["acc"] = int(m.group(1))
    # a function that is not a kernel runs inside some caller's allocation, which is not known here: any amount register of the
    # form 8 k + 7 is reported (the library's device functions are all inlined today, so this finds nothing)
    for k, lst in shifts.items():
        if k in nxt:
            continue
        for reg, text in lst:
            if reg % 8 == 7:
                found.append((k, text, reg, -1))
    for k, info in nxt.items():
        # arch VGPRs end at accum_offset when AGPRs follow, else at next_free_vgpr; the allocation is in blocks of 8
        arch = info.get("next", 0)
        if info.get("acc") and info["acc"] < arch: arch = info["acc"]
        for reg, text in shifts.get(k, []):
            if reg % 8 == 7 and reg + 1 >= arch:
                found.append((k, text, reg, arch))
    return found


def main(argv):
    files = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    for f in files:
        hits = scan(device_asm(f))
        for k, text, reg, arch in hits:
            print(f"{os.path.basename(f)}: {k}: `{text}` (amount in v{reg}, {arch if arch >= 0 else 'callee: unknown'} VGPRs)")
        bad += len(hits)
    print(f"{bad} 64-bit shift(s) with the amount in the last allocated VGPR")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
