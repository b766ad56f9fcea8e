"""Fingerprints of the compiled kernels inside libbwts_hip.so: sha256 over a kernel's machine code and its kernel descriptor
(without the descriptor's code-entry offset, which depends on where the linker put the OTHER kernels).

A PMC traffic record under profiles/ vouches for a KERNEL, not for the source file it lives in: bench.py and tools/pmc_summary.py
tie a record to this fingerprint, so a diagnostic edit elsewhere in the file (a host-side trace line, another kernel) leaves the
record valid, and any change that alters the kernel's code -- its own source, a device helper it inlines, a compiler flag --
retires it.

The .so carries one clang offload bundle per translation unit in .hip_fatbin; each bundle holds a gfx950 ELF code object.  Both
formats are read here directly (no external tool: the bench runs where only the repo snapshot exists).

    python tools/kernel_fingerprint.py [<lib.so>] [<name substring> ...]     # prints  <sha256-16>  <mangled symbol>
"""
import hashlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "bijective-bwt_amd", "libbwts_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    """Yields the device ELF images of every uncompressed offload bundle found in `blob`."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        (count,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        if count > 64:                   # not a bundle header (the magic inside some string table): move on
            pos += len(MAGIC)
            continue
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode(errors="replace")
            q += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def _elf_symbols(elf):
    """(name, section bytes-relative value, size, section index) of every symbol in .symtab of a little-endian ELF64 image, and the
    section table as (name, addr, offset, size)."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return [], []
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, addr, off, size, link, _info, _align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append({"name_off": name, "type": typ, "addr": addr, "off": off, "size": size, "link": link, "entsize": entsize})
    strtab = secs[shstrndx]

    def cstr(tab, o):
        s = tab["off"] + o
        return elf[s:elf.index(b"\0", s)].decode(errors="replace")

    for s in secs:
        s["name"] = cstr(strtab, s["name_off"])
    syms = []
    for s in secs:
        if s["type"] != 2:               # SHT_SYMTAB
            continue
        st = secs[s["link"]]
        for k in range(s["size"] // 24):
            name, info, _other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, s["off"] + 24 * k)
            syms.append((cstr(st, name), value, size, shndx, info & 15))
    return syms, secs


def kernel_fingerprints(lib_path=DEFAULT_LIB):
    """{mangled kernel symbol: sha256 hex of (code bytes + kernel descriptor bytes)} for every kernel of every code object."""
    blob = open(lib_path, "rb").read()
    out = {}
    for _triple, elf in _code_objects(blob):
        syms, secs = _elf_symbols(elf)
        by_name = {s[0]: s for s in syms}
        for name, value, size, shndx, typ in syms:
            if not name.endswith(".kd") or shndx == 0 or shndx >= len(secs):
                continue
            fn = by_name.get(name[:-3])
            if fn is None or fn[3] == 0 or fn[3] >= len(secs):
                continue
            h = hashlib.sha256()
            for v, sz, ndx, is_kd in ((fn[1], fn[2], fn[3], False), (value, size, shndx, True)):
                sec = secs[ndx]
                start = sec["off"] + (v - sec["addr"])
                part = elf[start:start + sz]
                if is_kd and len(part) >= 24:
                    # bytes 16..23 = kernel_code_entry_byte_offset: the distance from the descriptor to the code, which moves when
                    # ANOTHER kernel of the translation unit grows or one is added -- not a property of this kernel
                    part = part[:16] + b"\0" * 8 + part[24:]
                h.update(part)
            out[name[:-3]] = h.hexdigest()
    return out


def matching(fps, name_prefix, template_args=None):
    """Symbols of `fps` whose mangled name holds <len><name_prefix...> (Itanium: the unqualified name is length-prefixed), optionally
    narrowed to one instantiation by its mangled template-argument list (e.g. 'ILb0ELb0ELb1EE')."""
    hits = []
    for sym in fps:
        i = sym.find(name_prefix)
        if i <= 0 or not sym[i - 1].isdigit():
            continue
        if template_args is not None and (name_prefix + template_args) not in sym:
            continue
        hits.append(sym)
    return sorted(hits)


def pretty(sym):
    """name<template arguments> of a mangled kernel symbol, as far as rocprofv3's demangled names need it to be matched: the unqualified
    name and a template-argument list made of bool / int literals (all the library's kernel templates take are those and types).
    Returns (name, "<a, b, ...>" or "" when there is no list or it holds something else)."""
    import re
    m = re.match(r"_Z(\d+)", sym)
    if not m:
        return sym, ""
    ln = int(m.group(1))
    name = sym[m.end():m.end() + ln]
    rest = sym[m.end() + ln:]
    if not rest.startswith("I"):
        return name, ""
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        mm = re.match(r"L([bijlmxy])(n?)(\d+)E", rest[i:])
        if not mm:
            return name, ""
        v = mm.group(3)
        args.append(("true" if v != "0" else "false") if mm.group(1) == "b" else ("-" if mm.group(2) else "") + v + ("u" if mm.group(1) in "jmy" else ""))
        i += mm.end()
    return name, "<" + ", ".join(args) + ">"


def combined(fps, symbols):
    """One hash for a set of kernels (a record that sums several kernels' traffic)."""
    h = hashlib.sha256()
    for s in sorted(symbols):
        h.update(s.encode() + b"\0" + fps[s].encode() + b"\n")
    return h.hexdigest()


def main(argv):
    lib = argv[0] if argv and argv[0].endswith(".so") else DEFAULT_LIB
    subs = [a for a in argv if not a.endswith(".so")]
    fps = kernel_fingerprints(lib)
    for sym in sorted(fps):
        if not subs or any(s in sym for s in subs):
            print(fps[sym][:16], sym)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
