"""CPU suite: the C-ABI library loads and exports every symbol include/bwts.h declares; the CLIs
keep the reference's argument and error behaviour (no compute without a GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bijective-bwt_amd")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(os.path.join(PKG, "libbwts_hip.so")) or not os.path.exists(os.path.join(PKG, "mk_bwts")):
        subprocess.check_call(["make", "-C", PKG, "-j4", "all"], stdout=subprocess.DEVNULL)
    return PKG


def test_header_symbols_exported(built, pkg):
    L = pkg.lib()
    for name, listed in (("bwts.h", pkg.EXPORTS), ("bwts_test.h", pkg.TEST_EXPORTS)):
        header = open(os.path.join(ROOT, "include", name)).read()
        declared = sorted(set(re.findall(r"\b(bwts_[a-z0-9_]+)\s*\(", header)))
        assert declared, "no declarations parsed"
        missing = [s for s in declared if not hasattr(L, s)]
        assert not missing, missing
        assert sorted(listed) == declared
    # the drop-in header carries no harness or debug entry points
    assert not re.search(r"bwts_(debug|generate|device_alloc|copy_to)", open(os.path.join(ROOT, "include", "bwts.h")).read())


def test_strerror_and_names(pkg):
    L = pkg.lib()
    assert L.bwts_strerror(0) == b"ok"
    for code in range(-7, 0):
        assert L.bwts_strerror(code) and L.bwts_strerror(code) != b"unknown error"
    assert [L.bwts_kernel_class_name(i).decode() for i in range(pkg.K_COUNT)] == pkg.K_NAMES


def test_timings_struct_layout_matches_header(pkg):
    # bwts_timings: 3 doubles, 2 u64, 4 u32, 3 u64, the per-round counts, K_COUNT * (double + 3 u64), H_COUNT doubles, then 2 u32
    import ctypes
    assert ctypes.sizeof(pkg.KernelStat) == 32
    assert ctypes.sizeof(pkg.Timings) == 3 * 8 + 2 * 8 + 4 * 4 + 3 * 8 + pkg.MAX_ROUND_STATS * 8 + pkg.K_COUNT * 32 + pkg.H_COUNT * 8 + 8
    L = pkg.lib()
    L.bwts_host_cost_name.restype = ctypes.c_char_p
    assert [L.bwts_host_cost_name(i).decode() for i in range(pkg.H_COUNT)] == pkg.H_NAMES


def test_null_arguments_rejected(pkg):
    L = pkg.lib()
    assert L.bwts_ctx_create(None, 0) == -1
    assert L.bwts_forward(None, None, 0, None) == -1
    assert L.bwts_inverse_device(None, None, 5, None) == -1


def _run(args, **kw):
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_cli_usage(built):
    r = _run([os.path.join(built, "mk_bwts")])
    assert r.returncode == 1 and r.stdout == b""
    assert r.stderr.decode().splitlines() == ["Usage: mk_bwts_sa <infile> [<outfile.bwts>]",
                                              "If unspecified, output is written to standard output"]   # mk_bwts_sa.c:35-36
    r = _run([os.path.join(built, "unbwts")])
    assert r.returncode == 1
    assert r.stderr.decode().splitlines() == ["Usage: unbwts <infile.bwts> [<outfile>]",
                                              "If output file name is unspecified, a name is generated"]  # unbwts.c:22-23


def test_cli_input_errors(built, tmp_path):
    for prog in ("mk_bwts", "unbwts"):
        r = _run([os.path.join(built, prog), str(tmp_path / "missing")])
        assert r.returncode == 1 and b"No such file or directory" in r.stderr      # map_file.c:22-25
        empty = tmp_path / "empty"
        empty.write_bytes(b"")
        r = _run([os.path.join(built, prog), str(empty)])
        assert r.returncode == 1 and r.stderr.decode().strip() == "%s: Invalid argument" % empty   # map_file.c:36-40


def test_map_file_helper(built, tmp_path):
    """The retained helper API (map_file.h) keeps the reference's shapes: map_in divides by sizeof(*ptr)."""
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <stdio.h>
#include "map_file.h"
int main(int argc, char **argv) {
    unsigned int *w; long n; unsigned char *b; long nb;
    map_in(w, n, argv[1]);
    map_in(b, nb, argv[1]);
    ptr_range r = map_input_file(argv[1]);
    printf("%ld %ld %ld %u %d\n", n, nb, (long)((char*)r.ep - (char*)r.sp), w[1], b[0]);
    unmap_file(r);
    return 0;
}''')
    data = tmp_path / "d.bin"
    data.write_bytes(bytes(range(16)))
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src),
                           os.path.join(built, "cli", "map_file.c")])
    out = subprocess.check_output([str(exe), str(data)]).decode().split()
    assert out == ["4", "16", "16", str(0x07060504), "0"]


def test_no_cpu_fallback_without_gpu(pkg):
    """Without a GPU the product must fail loudly, never compute on the CPU."""
    import ctypes
    h = ctypes.c_void_p()
    rc = pkg.lib().bwts_ctx_create(ctypes.byref(h), 0)
    if rc == 0:       # a GPU is present (GPU box): nothing to check here
        pkg.lib().bwts_ctx_destroy(h)
        pytest.skip("GPU present")
    assert rc == -2
    with pytest.raises(pkg.BwtsError):
        pkg.mk_bwts(b"banana")


def test_product_does_not_touch_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may use oracle/."""
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_" not in text and "libbwts_oracle" not in text, os.path.join(dirpath, f)
