"""ctypes binding of the CPU oracle (oracle/libbwts_oracle.so) -- test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libbwts_oracle.so")
REF_UNBWTS = os.path.join(ORACLE_DIR, "_ref", "unbwts")

KINDS = {"uniform256": 0, "zipf": 1, "dna": 2, "text": 3}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libbwts_oracle.so"])
        L = ctypes.CDLL(ORACLE_SO)
        u8p = ctypes.c_void_p
        L.oracle_forward.argtypes = [u8p, ctypes.c_int64, u8p]
        L.oracle_forward.restype = ctypes.c_int
        L.oracle_forward64.argtypes = [u8p, ctypes.c_int64, u8p]
        L.oracle_forward64.restype = ctypes.c_int
        L.oracle_forward_timed.argtypes = [u8p, ctypes.c_int64, u8p, ctypes.POINTER(ctypes.c_double)]
        L.oracle_forward_timed.restype = ctypes.c_int
        L.oracle_forward_def.argtypes = [u8p, ctypes.c_int64, u8p]
        L.oracle_forward_def.restype = ctypes.c_int
        L.oracle_inverse.argtypes = [u8p, ctypes.c_int64, u8p]
        L.oracle_inverse.restype = ctypes.c_int
        L.oracle_suffix_array.argtypes = [u8p, ctypes.c_void_p, ctypes.c_int64]
        L.oracle_suffix_array.restype = ctypes.c_int
        L.oracle_lyndon_starts.argtypes = [u8p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
        L.oracle_lyndon_starts.restype = ctypes.c_int64
        L.oracle_generate.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, u8p]
        L.oracle_generate.restype = None
        _lib = L
    return _lib


def _as_u8(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


def _call(fn, data):
    a = _as_u8(data)
    out = np.empty_like(a)
    rc = fn(a.ctypes.data, a.size, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle call failed rc=%d" % rc)
    return out


def forward(data):
    return _call(lib().oracle_forward, data)


def forward64(data):
    """The pipeline's 64-bit-index instance (what forward() switches to from 2^31 - 1 bytes on)."""
    return _call(lib().oracle_forward64, data)


def forward_def(data):
    return _call(lib().oracle_forward_def, data)


def inverse(data):
    return _call(lib().oracle_inverse, data)


def forward_timed(data):
    a = _as_u8(data)
    out = np.empty_like(a)
    ph = (ctypes.c_double * 4)()
    rc = lib().oracle_forward_timed(a.ctypes.data, a.size, out.ctypes.data, ph)
    if rc != 0:
        raise RuntimeError("oracle call failed rc=%d" % rc)
    return out, list(ph)


def suffix_array(data):
    a = _as_u8(data)
    sa = np.empty(a.size, dtype=np.int32)
    rc = lib().oracle_suffix_array(a.ctypes.data, sa.ctypes.data, a.size)
    if rc != 0:
        raise RuntimeError("oracle call failed rc=%d" % rc)
    return sa


def lyndon_starts(data):
    a = _as_u8(data)
    st = np.empty(max(a.size, 1), dtype=np.int64)
    k = lib().oracle_lyndon_starts(a.ctypes.data, a.size, st.ctypes.data, st.size)
    return st[:k].copy()


def generate(kind, n, seed, off=0):
    out = np.empty(n, dtype=np.uint8)
    lib().oracle_generate(KINDS[kind], seed, off, n, out.ctypes.data)
    return out


def have_ref_unbwts():
    return os.path.exists(REF_UNBWTS) and os.access(REF_UNBWTS, os.X_OK)


def ref_unbwts(data, tmpdir):
    """Run the reference's own inverse program (built from its untouched sources)."""
    a = _as_u8(data)
    src = os.path.join(tmpdir, "in.bwts")
    dst = os.path.join(tmpdir, "out.raw")
    a.tofile(src)
    subprocess.check_call([REF_UNBWTS, src, dst])
    return np.fromfile(dst, dtype=np.uint8)
