"""Host-side Python mirror of the MI355X bijective-BWT engine.

Thin ctypes binding of the C-ABI in ``include/bwts.h`` (``libbwts_hip.so``).  The two
module-level functions keep the reference programs' meaning:

* :func:`mk_bwts`  -- forward BWTS of a byte string  (``/root/reference/mk_bwts_sa.c:33-65``)
* :func:`unbwts`   -- inverse BWTS of a byte string  (``/root/reference/unbwts.c:19-92``)

There is no CPU path here: if the HIP library is missing, or no GPU is usable, calls raise
:class:`BwtsError`.  (The CPU oracle lives under ``oracle/`` and is test infrastructure.)

The directory name contains a hyphen, so import it through ``load_package()`` of the
repo-root ``__graft_entry__`` or ``importlib`` (see ``tests/conftest.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# BWTS_LIB_OVERRIDE: another build of the same library (A/B timing sessions on one GPU box, tools/sessions/*.sh); never set in tests
LIB_PATH = os.environ.get("BWTS_LIB_OVERRIDE") or os.path.join(PKG_DIR, "libbwts_hip.so")

KINDS = {"uniform256": 0, "zipf": 1, "dna": 2, "text": 3}

K_NAMES = ["histogram", "keybuild", "radix_hist", "radix_scan", "radix_scatter", "rerank", "lyndon", "emit",
           "lf_build", "walk", "listrank", "walk_emit", "other", "radix_scatter_main", "round"]
K_COUNT = len(K_NAMES)
MAX_ROUND_STATS = 40
H_NAMES = ["init", "module_load", "io_alloc", "staging_alloc", "arena_alloc"]
H_COUNT = len(H_NAMES)

# every symbol include/bwts.h declares (the drop-in surface) ...
EXPORTS = [
    "bwts_ctx_create", "bwts_ctx_destroy", "bwts_ctx_release_memory", "bwts_forward", "bwts_inverse", "bwts_forward_sink", "bwts_inverse_sink",
    "bwts_forward_device", "bwts_inverse_device", "bwts_last_timings", "bwts_kernel_class_name", "bwts_strerror",
    "bwts_last_hip_error", "bwts_set_timing", "bwts_host_alloc", "bwts_host_free", "bwts_host_cost_name",
    "bwts_forward_batch", "bwts_inverse_batch",
]
# ... and include/bwts_test.h (harness and unit-test hooks)
TEST_EXPORTS = [
    "bwts_generate_device", "bwts_device_alloc", "bwts_device_free", "bwts_copy_to_device", "bwts_copy_to_host",
    "bwts_device_equal", "bwts_debug_sort_pairs", "bwts_debug_suffix_array", "bwts_debug_lyndon",
]


class BwtsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bwts error %d: %s" % (code, msg))
        self.code = code


class KernelStat(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double), ("launches", ctypes.c_uint64), ("elems", ctypes.c_uint64),
                ("alg_bytes", ctypes.c_uint64)]


class Timings(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double), ("h2d_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double),
                ("n", ctypes.c_uint64), ("factors", ctypes.c_uint64), ("rounds", ctypes.c_uint32),
                ("lyndon_rounds", ctypes.c_uint32), ("key_symbols", ctypes.c_uint32), ("key_bits", ctypes.c_uint32),
                ("active_after_round0", ctypes.c_uint64), ("unvisited", ctypes.c_uint64), ("device_bytes", ctypes.c_uint64),
                ("round_active", ctypes.c_uint64 * MAX_ROUND_STATS), ("k", KernelStat * K_COUNT),
                ("host_ms", ctypes.c_double * H_COUNT), ("attempts", ctypes.c_uint32), ("reserved_", ctypes.c_uint32)]

    def as_dict(self):
        d = {f: getattr(self, f) for f, _ in self._fields_ if f not in ("k", "round_active", "host_ms")}
        d["host_ms"] = {H_NAMES[i]: float(self.host_ms[i]) for i in range(H_COUNT)}
        d["round_active"] = [int(v) for v in self.round_active[: max(int(self.rounds), 1)]]
        d["kernels"] = {K_NAMES[i]: {"ms": self.k[i].ms, "launches": self.k[i].launches, "elems": self.k[i].elems,
                                     "alg_bytes": self.k[i].alg_bytes} for i in range(K_COUNT) if self.k[i].launches}
        return d


SINK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)

_lib = None


def build(verbose=False):
    """Compile libbwts_hip.so (gfx950) and the CLIs in-tree with hipcc."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", PKG_DIR, "-j4", "all"], stdout=out)


def lib():
    """The loaded C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BwtsError(-2, "libbwts_hip.so is not built (run `make -C bijective-bwt_amd` or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        vp, u64, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
        L.bwts_ctx_create.argtypes = [ctypes.POINTER(vp), i32]
        L.bwts_ctx_destroy.argtypes = [vp]
        L.bwts_ctx_destroy.restype = None
        L.bwts_ctx_release_memory.argtypes = [vp]
        L.bwts_ctx_release_memory.restype = ctypes.c_int
        for name in ("bwts_forward", "bwts_inverse", "bwts_forward_device", "bwts_inverse_device"):
            getattr(L, name).argtypes = [vp, vp, u64, vp]
        L.bwts_last_timings.argtypes = [vp, ctypes.POINTER(Timings)]
        L.bwts_kernel_class_name.argtypes = [i32]
        L.bwts_kernel_class_name.restype = ctypes.c_char_p
        L.bwts_strerror.argtypes = [i32]
        L.bwts_strerror.restype = ctypes.c_char_p
        L.bwts_last_hip_error.argtypes = [vp]
        L.bwts_set_timing.argtypes = [vp, i32]
        L.bwts_host_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
        L.bwts_host_free.argtypes = [vp, vp]
        for name in ("bwts_forward_batch", "bwts_inverse_batch"):
            getattr(L, name).argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(u64), ctypes.POINTER(vp)]
        for name in ("bwts_forward_sink", "bwts_inverse_sink"):
            getattr(L, name).argtypes = [vp, vp, u64, SINK_FN, vp]
        L.bwts_generate_device.argtypes = [vp, i32, u64, u64, vp]
        L.bwts_device_alloc.argtypes = [vp, u64, ctypes.POINTER(vp)]
        L.bwts_device_free.argtypes = [vp, vp]
        L.bwts_copy_to_device.argtypes = [vp, vp, vp, u64]
        L.bwts_copy_to_host.argtypes = [vp, vp, vp, u64]
        L.bwts_device_equal.argtypes = [vp, vp, vp, u64, ctypes.POINTER(i32)]
        L.bwts_debug_sort_pairs.argtypes = [vp, vp, vp, u64, i32]
        L.bwts_debug_suffix_array.argtypes = [vp, vp, u64, vp]
        L.bwts_debug_lyndon.argtypes = [vp, vp, u64, vp, u64, ctypes.POINTER(u64)]
        _lib = L
    return _lib


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


class DeviceBuffer:
    """A raw device allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = ctypes.c_void_p()
        ctx._check(lib().bwts_device_alloc(ctx._h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            lib().bwts_device_free(self.ctx._h, self.ptr)
            self.ptr = None

    def upload(self, data):
        a = _u8(data)
        assert a.size <= self.nbytes
        self.ctx._check(lib().bwts_copy_to_device(self.ctx._h, self.ptr, a.ctypes.data, a.size))

    def download(self, nbytes=None):
        n = self.nbytes if nbytes is None else int(nbytes)
        out = np.empty(n, dtype=np.uint8)
        self.ctx._check(lib().bwts_copy_to_host(self.ctx._h, out.ctypes.data, self.ptr, n))
        return out


class Context:
    """One GPU context (``bwts_ctx``): owns a stream, a device arena and pinned staging."""

    def __init__(self, device=0):
        self._h = None
        h = ctypes.c_void_p()
        rc = lib().bwts_ctx_create(ctypes.byref(h), int(device))
        if rc != 0:
            raise BwtsError(rc, lib().bwts_strerror(rc).decode())
        self._h = h
        self.device = int(device)

    def close(self):
        if self._h:
            lib().bwts_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_memory(self):
        """Hands the device memory the context keeps between calls back to the device (bwts_ctx_release_memory)."""
        self._check(lib().bwts_ctx_release_memory(self._h))

    def _check(self, rc):
        if rc != 0:
            msg = lib().bwts_strerror(rc).decode()
            if rc == -4:
                msg += " (hipError %d)" % lib().bwts_last_hip_error(self._h)
            raise BwtsError(rc, msg)

    # -- host-buffer transforms ------------------------------------------------------
    def _host(self, fn, data):
        a = _u8(data)
        if a.size == 0:
            raise BwtsError(-1, "empty input (the reference fails on empty files: map_file.c:36-40)")
        out = np.empty_like(a)
        self._check(fn(self._h, a.ctypes.data, a.size, out.ctypes.data))
        return out

    def forward(self, data):
        return self._host(lib().bwts_forward, data)

    def inverse(self, data):
        return self._host(lib().bwts_inverse, data)

    # -- device-buffer transforms ------------------------------------------------------
    def forward_device(self, d_in, n, d_out):
        self._check(lib().bwts_forward_device(self._h, _ptr(d_in), int(n), _ptr(d_out)))

    def inverse_device(self, d_in, n, d_out):
        self._check(lib().bwts_inverse_device(self._h, _ptr(d_in), int(n), _ptr(d_out)))

    def set_timing(self, level=2):
        """HIP-event times in timings(): 0 off (default), 1 dominant kernels only, 2 every class (~1 ms per call at 1 GiB)."""
        self._check(lib().bwts_set_timing(self._h, int(level)))

    def _sink(self, fn, data):
        a = _u8(data)
        pieces = []

        def take(_user, ptr, length):
            pieces.append(ctypes.string_at(ptr, length))
            return 0

        cb = SINK_FN(take)
        self._check(fn(self._h, a.ctypes.data, a.size, cb, None))
        return np.frombuffer(b"".join(pieces), dtype=np.uint8)

    def forward_sink(self, data):
        """forward() with the output delivered through the sink callback (what the CLIs use)."""
        return self._sink(lib().bwts_forward_sink, data)

    def inverse_sink(self, data):
        return self._sink(lib().bwts_inverse_sink, data)

    def host_alloc(self, nbytes):
        """A pinned host block as a numpy uint8 array (freed with the context or host_free())."""
        p = ctypes.c_void_p()
        self._check(lib().bwts_host_alloc(self._h, int(nbytes), ctypes.byref(p)))
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(nbytes),))
        return arr, p.value

    def host_free(self, ptr):
        self._check(lib().bwts_host_free(self._h, ptr))

    def _batch(self, fn, arrays):
        arrs = [_u8(a) for a in arrays]
        outs = [np.empty_like(a) for a in arrs]
        k = len(arrs)
        ins_p = (ctypes.c_void_p * k)(*[a.ctypes.data for a in arrs])
        outs_p = (ctypes.c_void_p * k)(*[o.ctypes.data for o in outs])
        ns = (ctypes.c_uint64 * k)(*[a.size for a in arrs])
        self._check(fn(self._h, k, ins_p, ns, outs_p))
        return outs

    def forward_batch(self, arrays):
        """bwts_forward_batch: the transforms of several inputs, copies overlapped with the neighbours' transforms."""
        return self._batch(lib().bwts_forward_batch, arrays)

    def inverse_batch(self, arrays):
        return self._batch(lib().bwts_inverse_batch, arrays)

    def forward_into(self, a, out):
        """bwts_forward on caller-provided numpy buffers (no allocation inside the call)."""
        self._check(lib().bwts_forward(self._h, a.ctypes.data, a.size, out.ctypes.data))

    def inverse_into(self, a, out):
        self._check(lib().bwts_inverse(self._h, a.ctypes.data, a.size, out.ctypes.data))

    def timings(self):
        t = Timings()
        self._check(lib().bwts_last_timings(self._h, ctypes.byref(t)))
        return t

    # -- harness utilities ---------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def generate(self, kind, seed, n, d_out):
        self._check(lib().bwts_generate_device(self._h, KINDS[kind], int(seed), int(n), _ptr(d_out)))

    def device_equal(self, d_a, d_b, nbytes):
        eq = ctypes.c_int(0)
        self._check(lib().bwts_device_equal(self._h, _ptr(d_a), _ptr(d_b), int(nbytes), ctypes.byref(eq)))
        return bool(eq.value)

    # -- unit-test hooks -------------------------------------------------------------------
    def debug_sort_pairs(self, keys, vals, key_bits=64):
        k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self._check(lib().bwts_debug_sort_pairs(self._h, k.ctypes.data, v.ctypes.data, k.size, int(key_bits)))
        return k, v

    def debug_suffix_array(self, data):
        a = _u8(data)
        sa = np.empty(a.size, dtype=np.uint32)
        self._check(lib().bwts_debug_suffix_array(self._h, a.ctypes.data, a.size, sa.ctypes.data))
        return sa

    def debug_lyndon(self, data):
        a = _u8(data)
        st = np.empty(a.size, dtype=np.uint64)
        cnt = ctypes.c_uint64(0)
        self._check(lib().bwts_debug_lyndon(self._h, a.ctypes.data, a.size, st.ctypes.data, st.size, ctypes.byref(cnt)))
        return st[: cnt.value].copy()


def _ptr(x):
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):      # torch tensor on the context's GPU
        return x.data_ptr()
    return int(x)


def mk_bwts(data, device=0):
    """Forward bijective BWT of ``data`` (bytes-like) -> numpy uint8 array of the same length."""
    with Context(device) as ctx:
        return ctx.forward(data)


def unbwts(data, device=0):
    """Inverse bijective BWT of ``data`` (bytes-like) -> numpy uint8 array of the same length."""
    with Context(device) as ctx:
        return ctx.inverse(data)
