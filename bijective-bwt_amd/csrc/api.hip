// api.hip -- the C-ABI of include/bwts.h: context, arenas, staging, timing, test hooks.
#include "internal.h"
#include "../../include/bwts_test.h"

#include <new>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

double wall_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

void bwts_trace_error(const char *file, int line, int rc)
{
    static const bool on = [] { const char *e = getenv("BWTS_TRACE_ERRORS"); return e && e[0] == '1'; }();
    if (on) fprintf(stderr, "[bwts] %s:%d: rc %d\n", file, line, rc);
}

// ------------------------------------------------------------------------------------
// environment knobs
// ------------------------------------------------------------------------------------
// Read once per context.  Always: diagnostics and the staging tuning below.  Everything else selects alternate code paths that exist
// for the test suite (tests/test_gpu_parity.py::test_alternate_paths) and for A/B timing sessions: those are only looked at when
// BWTS_TEST_KNOBS=1 is set, so a process that merely inherits a stray BWTS_* variable runs the product paths.
static const char *const kPublicKnobs[] = {"BWTS_TIMINGS", "BWTS_TRACE_ERRORS", "BWTS_ROUND_TRACE", "BWTS_INV_TRACE", "BWTS_BATCH_TRACE",
                                           "BWTS_COPY_THREADS", "BWTS_H2D", "BWTS_D2H", "BWTS_D2H_SPLIT"};
extern char **environ;
int radix_config_count(void);

static void read_knobs(bwts_ctx *ctx)
{
    const char *gate = getenv("BWTS_TEST_KNOBS");
    const bool all = gate && gate[0] == '1';
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "BWTS_", 5) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (!eq) continue;
        std::string name(*e, (size_t)(eq - *e));
        bool pub = false;
        for (const char *k : kPublicKnobs) if (name == k) pub = true;
        if (pub || all) ctx->knobs.emplace_back(name, std::string(eq + 1));
    }
    ctx->rx_config = 0;
    if (const char *v = bwts_knob(ctx, "BWTS_RX_CONFIG")) { const int c = atoi(v); if (c >= 0 && c < radix_config_count()) ctx->rx_config = c; }
}

const char *bwts_knob(const bwts_ctx *ctx, const char *name)
{
    for (const auto &kv : ctx->knobs) if (kv.first == name) return kv.second.c_str();
    return nullptr;
}

// ------------------------------------------------------------------------------------
// arenas
// ------------------------------------------------------------------------------------
// BWTS_POISON=1 (a test switch): every block handed to a transform is filled with 0xA5 first, so that a kernel which reads memory
// nothing has written yet does so reproducibly -- whatever an earlier call or process left there -- instead of once in a blue moon
static bool poison_on(const bwts_ctx *ctx) { const char *e = bwts_knob(ctx, "BWTS_POISON"); return e && e[0] == '1'; }

// device blocks of the context, with guard bands when BWTS_GUARD=1
#define GUARD_BYTE 0x5C
#define GUARD_FREED 0x5D
// BWTS_TRACE_ALLOC=1: every block the context takes or gives up, with its address range, on stderr -- the map a GPU memory fault's
// address is read against
static void trace_alloc(const bwts_ctx *ctx, const char *what, const char *name, const void *p, size_t bytes)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("BWTS_TRACE_ALLOC"); on = (e && e[0] == '1') ? 1 : 0; }
    if (on) fprintf(stderr, "[bwts alloc] ctx %p %s %-16s [%p, %p) %zu bytes\n", (const void *)ctx, what, name, p, (const void *)((const char *)p + bytes), bytes);
}
static hipError_t ctx_malloc(bwts_ctx *ctx, void **out, size_t bytes, const char *name)
{
    if (!ctx->guard) { const hipError_t e0 = hipMalloc(out, bytes); if (e0 == hipSuccess) trace_alloc(ctx, "device +", name, *out, bytes); return e0; }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes + 2 * ctx->guard);
    if (e != hipSuccess) return e;
    e = hipMemset(p, GUARD_BYTE, ctx->guard);
    if (e == hipSuccess) e = hipMemset((char *)p + ctx->guard + bytes, GUARD_BYTE, ctx->guard);
    if (e != hipSuccess) { (void)hipFree(p); return e; }
    *out = (char *)p + ctx->guard;
    ctx->guard_blocks.push_back({(char *)*out, bytes, name});
    return hipSuccess;
}
static hipError_t ctx_free(bwts_ctx *ctx, void *user)
{
    trace_alloc(ctx, "device -", "", user, 0);
    if (!ctx->guard) return hipFree(user);
    for (size_t i = 0; i < ctx->guard_blocks.size(); i++)
        if (ctx->guard_blocks[i].user == (char *)user) {
            const bwts_ctx::GuardBlock b = ctx->guard_blocks[i];
            ctx->guard_blocks.erase(ctx->guard_blocks.begin() + (long)i);
            // blocks of up to 64 MiB are not handed back: they stay mapped, filled with a pattern that guard_check() looks at
            if (b.bytes <= ((size_t)64 << 20) && hipDeviceSynchronize() == hipSuccess &&
                hipMemset((char *)user - ctx->guard, GUARD_FREED, b.bytes + 2 * ctx->guard) == hipSuccess) {
                ctx->guard_freed.push_back(b);
                return hipSuccess;
            }
            break;
        }
    return hipFree((char *)user - ctx->guard);
}
// after a transform: are all guard bands intact?
static int guard_check(bwts_ctx *ctx, const char *what)
{
    if (!ctx->guard) return BWTS_OK;
    std::vector<unsigned char> h(ctx->guard);
    int bad = 0;
    for (const auto &b : ctx->guard_blocks)
        for (int side = 0; side < 2; side++) {
            const char *src = side == 0 ? b.user - ctx->guard : b.user + b.bytes;
            HIPC(hipMemcpy(h.data(), src, ctx->guard, hipMemcpyDeviceToHost));
            size_t first = ctx->guard, last = 0, count = 0;
            for (size_t i = 0; i < ctx->guard; i++)
                if (h[i] != GUARD_BYTE) { if (first == ctx->guard) first = i; last = i; count++; }
            if (count) {
                bad++;
                fprintf(stderr, "[bwts guard] %s: block '%s' (%zu bytes): %zu byte(s) written %s it, offsets %ld .. %ld relative to the block's %s; first bytes:", what,
                        b.name, b.bytes, count, side == 0 ? "IN FRONT OF" : "BEHIND", side == 0 ? (long)first - (long)ctx->guard : (long)first,
                        side == 0 ? (long)last - (long)ctx->guard : (long)last, side == 0 ? "start" : "end");
                for (size_t i = first; i < first + 16 && i < ctx->guard; i++) fprintf(stderr, " %02x", h[i]);
                fprintf(stderr, "\n");
                HIPC(hipMemset((void *)src, GUARD_BYTE, ctx->guard));
            }
        }
    for (const auto &b : ctx->guard_freed) {
        const size_t total = b.bytes + 2 * ctx->guard;
        std::vector<unsigned char> f(total);
        HIPC(hipMemcpy(f.data(), b.user - ctx->guard, total, hipMemcpyDeviceToHost));
        size_t first = total, last = 0, count = 0;
        for (size_t i = 0; i < total; i++)
            if (f[i] != GUARD_FREED) { if (first == total) first = i; last = i; count++; }
        if (count) {
            bad++;
            fprintf(stderr, "[bwts guard] %s: GIVEN-UP block '%s' (%zu bytes) was written after the context let go of it: %zu byte(s), offsets %ld .. %ld from its start; first bytes:",
                    what, b.name, b.bytes, count, (long)first - (long)ctx->guard, (long)last - (long)ctx->guard);
            for (size_t i = first; i < first + 16 && i < total; i++) fprintf(stderr, " %02x", f[i]);
            fprintf(stderr, "\n");
            HIPC(hipMemset(b.user - ctx->guard, GUARD_FREED, total));
        }
    }
    return bad ? BWTS_E_INTERNAL : BWTS_OK;
}

void bwts_stage_mark(bwts_ctx *ctx, const char *name)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("BWTS_STAGE_TRACE"); on = (e && e[0] == '1') ? 1 : 0; }
    if (!on) return;
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    fprintf(stderr, "[bwts stage] %s %s\n", name, e == hipSuccess ? "ok" : "FAILED");
    fflush(stderr);
}

// The arena changes hands on the thread that owns the context, and only with the context's stream drained: nothing that was
// enqueued can still use the block that is given up.
int arena_release(bwts_ctx *ctx)
{
    if (!ctx->arena) return BWTS_OK;
    HIPC(hipStreamSynchronize(ctx->stream));
    HIPC(ctx_free(ctx, ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_cap = 0;
    ctx->arena_off = 0;
    return BWTS_OK;
}

void arena_install(bwts_ctx *ctx, void *block, size_t bytes, double alloc_ms)
{
    trace_alloc(ctx, "device +", "arena", block, bytes);
    ctx->arena = (char *)block;
    ctx->arena_cap = bytes;
    ctx->arena_off = 0;
    ctx->host_ms[BWTS_H_ARENA_ALLOC] += alloc_ms;
}

int arena_reserve(bwts_ctx *ctx, size_t bytes)
{
    bytes = align_up(bytes, 1 << 20);
    if (bytes > ctx->arena_cap) {
        const double t0 = wall_ms();
        BWTS_TRY(arena_release(ctx));
        void *p = nullptr;
        if (ctx_malloc(ctx, &p, bytes, "arena") != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
        ctx->arena = (char *)p;
        ctx->arena_cap = bytes;
        ctx->host_ms[BWTS_H_ARENA_ALLOC] += wall_ms() - t0;
    }
    ctx->arena_off = 0;
    if (ctx->arena && poison_on(ctx)) HIPC(hipMemsetAsync(ctx->arena, 0xA5, ctx->arena_cap, ctx->stream));
    return BWTS_OK;
}

void arena_reset(bwts_ctx *ctx) { ctx->arena_off = 0; }

void *arena_alloc(bwts_ctx *ctx, size_t bytes)
{
    bytes = align_up(bytes ? bytes : 1, 256);
    if (ctx->arena_off + bytes > ctx->arena_cap) return nullptr;
    void *p = ctx->arena + ctx->arena_off;
    trace_alloc(ctx, "  arena:", "array", p, bytes);
    ctx->arena_off += bytes;
    return p;
}

int aux_reserve_slot(bwts_ctx *ctx, int slot, size_t bytes, char **base)
{
    bytes = align_up(bytes, 1 << 20);
    if (bytes > ctx->aux_cap[slot]) {
        const double t0 = wall_ms();
        // contents of a previous, smaller block are never live across this call
        if (ctx->aux[slot]) { HIPC(hipStreamSynchronize(ctx->stream)); HIPC(ctx_free(ctx, ctx->aux[slot])); ctx->aux[slot] = nullptr; ctx->aux_cap[slot] = 0; }
        void *p = nullptr;
        static const char *const aux_names[BWTS_AUX_SLOTS] = {"side block 0", "side block 1", "side block 2", "side block 3", "side block 4"};
        if (ctx_malloc(ctx, &p, bytes, aux_names[slot]) != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
        ctx->aux[slot] = (char *)p;
        ctx->aux_cap[slot] = bytes;
        ctx->host_ms[BWTS_H_ARENA_ALLOC] += wall_ms() - t0;
    }
    *base = ctx->aux[slot];
    if (ctx->aux[slot] && poison_on(ctx)) HIPC(hipMemsetAsync(ctx->aux[slot], 0xA5, ctx->aux_cap[slot], ctx->stream));
    return BWTS_OK;
}

int ensure_dyn_lds(bwts_ctx *ctx, const void *kernel, size_t bytes)
{
    for (const void *k : ctx->lds_granted) if (k == kernel) return BWTS_OK;
    HIPC(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->lds_granted.push_back(kernel);
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// timing spans
// ------------------------------------------------------------------------------------
static hipEvent_t take_event(bwts_ctx *ctx)
{
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[ctx->ev_used++];
}

void spans_reset(bwts_ctx *ctx)
{
    ctx->ev_used = 0;
    ctx->spans.clear();
    memset(&ctx->tm, 0, sizeof(ctx->tm));
}

int span_begin(bwts_ctx *ctx, int cls, u64 elems, u64 alg_bytes)
{
    ctx->tm.k[cls].launches++;
    ctx->tm.k[cls].elems += elems;
    ctx->tm.k[cls].alg_bytes += alg_bytes;
    if (ctx->timing < 2 && !(ctx->timing == 1 && (cls == BWTS_K_RADIX_SCATTER_MAIN || cls == BWTS_K_WALK || cls == BWTS_K_ROUND))) return -1;
    TimedSpan sp;
    sp.cls = cls;
    sp.a = take_event(ctx);
    sp.b = take_event(ctx);
    if (!sp.a || !sp.b) return -1;
    (void)hipEventRecord(sp.a, ctx->stream);
    ctx->spans.push_back(sp);
    return (int)ctx->spans.size() - 1;
}

void span_end(bwts_ctx *ctx, int span)
{
    if (span < 0 || (size_t)span >= ctx->spans.size()) return;
    (void)hipEventRecord(ctx->spans[span].b, ctx->stream);
}

int spans_resolve(bwts_ctx *ctx)
{
    HIPC(hipStreamSynchronize(ctx->stream));
    for (const TimedSpan &sp : ctx->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) ctx->tm.k[sp.cls].ms += ms;
    }
    return BWTS_OK;
}

int read_small(bwts_ctx *ctx, int first, int count)
{
    HIPC(hipMemcpyAsync(ctx->h_small + first, ctx->d_small + first, (size_t)count * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
extern "C" int bwts_ctx_create(bwts_ctx **out, int device_id)
{
    if (!out) return BWTS_E_ARG;
    *out = nullptr;
    const double t_create = wall_ms();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); return BWTS_E_NODEVICE; }
    if (device_id < 0 || device_id >= count) return BWTS_E_NODEVICE;
    bwts_ctx *ctx = new (std::nothrow) bwts_ctx();
    if (!ctx) return BWTS_E_NOMEM;
    ctx->device = device_id;
    read_knobs(ctx);
    { const char *e = bwts_knob(ctx, "BWTS_TIMINGS"); ctx->timing = (e && e[0] == '1') ? 2 : 0; }
    { const char *e = bwts_knob(ctx, "BWTS_GUARD"); const int mib = e ? atoi(e) : 0; ctx->guard = mib >= 1 && mib <= 256 ? ((size_t)mib << 20) : 0; }     // MiB per band
    if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return BWTS_E_NODEVICE; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return BWTS_E_HIP; }
    void *p = nullptr;
    if (ctx_malloc(ctx, &p, 4096 * sizeof(u64), "small words") != hipSuccess) { bwts_ctx_destroy(ctx); return BWTS_E_NOMEM; }
    ctx->d_small = (u64 *)p;
    if (hipMemset(p, 0, 4096 * sizeof(u64)) != hipSuccess) { bwts_ctx_destroy(ctx); return BWTS_E_HIP; }     // (counters that kernels expect at zero)
    if (hipHostMalloc(&p, 4096 * sizeof(u64), hipHostMallocDefault) != hipSuccess) { bwts_ctx_destroy(ctx); return BWTS_E_NOMEM; }
    trace_alloc(ctx, "pinned +", "small words", p, 4096 * sizeof(u64));
    ctx->h_small = (u64 *)p;
    if (hipEventCreate(&ctx->ev_begin) != hipSuccess || hipEventCreate(&ctx->ev_end) != hipSuccess) { bwts_ctx_destroy(ctx); return BWTS_E_HIP; }
    ctx->host_ms[BWTS_H_INIT] = wall_ms() - t_create;
    *out = ctx;
    return BWTS_OK;
}

extern "C" void bwts_ctx_destroy(bwts_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->arena) (void)ctx_free(ctx, ctx->arena);
    for (int i = 0; i < BWTS_AUX_SLOTS; i++) if (ctx->aux[i]) (void)ctx_free(ctx, ctx->aux[i]);
    for (char *b : ctx->tied_blk) (void)hipFree(b);
    if (ctx->d_small) (void)ctx_free(ctx, ctx->d_small);
    for (const auto &b : ctx->guard_freed) (void)hipFree(b.user - ctx->guard);
    ctx->guard_freed.clear();
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    for (Stager &sg : ctx->stg) {
        if (sg.pool) { sg.pool->shutdown(); delete sg.pool; }
        for (int i = 0; i < STAGE_SLOTS; i++) {
            if (sg.pinned[i]) (void)hipHostFree(sg.pinned[i]);
            if (sg.slot_ev[i]) (void)hipEventDestroy(sg.slot_ev[i]);
            if (sg.slot_ev2[i]) (void)hipEventDestroy(sg.slot_ev2[i]);
        }
        if (sg.copy_stream) (void)hipStreamDestroy(sg.copy_stream);
        if (sg.own_stream && sg.stream) (void)hipStreamDestroy(sg.stream);
    }
    for (int i = 0; i < 4; i++) if (ctx->d_io[i]) (void)ctx_free(ctx, ctx->d_io[i]);
    for (const auto &b : ctx->host_blocks) (void)hipHostFree(b.first);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// A context keeps what its last calls allocated on the device (a 12 GiB input leaves ~100-200 GiB there) so that the next call starts at
// once; a caller who wants the memory back without giving up the context asks for it here.  Pinned staging and the small blocks stay.
extern "C" int bwts_ctx_release_memory(bwts_ctx *ctx)
{
    if (!ctx) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    HIPC(hipStreamSynchronize(ctx->stream));
    BWTS_TRY(arena_release(ctx));
    for (int i = 0; i < BWTS_AUX_SLOTS; i++)
        if (ctx->aux[i]) { HIPC(ctx_free(ctx, ctx->aux[i])); ctx->aux[i] = nullptr; ctx->aux_cap[i] = 0; }
    for (char *b : ctx->tied_blk) HIPC(hipFree(b));
    ctx->tied_blk.clear();
    for (int i = 0; i < 4; i++)
        if (ctx->d_io[i]) { HIPC(ctx_free(ctx, ctx->d_io[i])); ctx->d_io[i] = nullptr; ctx->d_io_cap[i] = 0; }
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// transforms
// ------------------------------------------------------------------------------------
typedef int (*device_impl_fn)(bwts_ctx *, const u8 *, u64, u8 *);
__global__ void pcie_copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, u64 vecs, u8 *__restrict__ dst_tail, const u8 *__restrict__ src_tail, u32 tail);

static int run_device(bwts_ctx *ctx, device_impl_fn fn, const void *d_in, u64 n, void *d_out)
{
    if (!ctx || !d_in || !d_out || n == 0) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    spans_reset(ctx);
    ctx->tm.n = n;
    ctx->call_block_bytes = 0;
    if (!ctx->launched) {
        // the library's code object is loaded at the first launch: pay (and account for) that outside the timed transform
        const double t0 = wall_ms();
        pcie_copy_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(nullptr, nullptr, 0, nullptr, nullptr, 0);
        HIPC(hipStreamSynchronize(ctx->stream));
        ctx->host_ms[BWTS_H_MODULE] += wall_ms() - t0;
        ctx->launched = true;
    }
    HIPC(hipEventRecord(ctx->ev_begin, ctx->stream));
    int rc = fn(ctx, (const u8 *)d_in, n, (u8 *)d_out);
    if (ctx->guard) {
        (void)hipStreamSynchronize(ctx->stream);
        const int grc = guard_check(ctx, fn == forward_device_impl ? "forward" : "inverse");
        if (rc == BWTS_OK) rc = grc;
    }
    if (rc != BWTS_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    HIPC(hipEventRecord(ctx->ev_end, ctx->stream));
    BWTS_TRY(spans_resolve(ctx));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
    ctx->tm.total_ms = ms;
    ctx->tm.device_bytes = ctx->arena_cap + ctx->d_io_cap[0] + ctx->d_io_cap[1] + ctx->call_block_bytes;
    for (int i = 0; i < BWTS_AUX_SLOTS; i++) ctx->tm.device_bytes += ctx->aux_cap[i];
    ctx->tm.device_bytes += ctx->d_io_cap[2] + ctx->d_io_cap[3];
    ctx->tm.device_bytes += ctx->tied_blk.size() * ((size_t)16 << ctx->tied_blk_lg);
    return BWTS_OK;
}

extern "C" int bwts_forward_device(bwts_ctx *ctx, const void *d_in, uint64_t n, void *d_out)
{
    return run_device(ctx, forward_device_impl, d_in, n, d_out);
}

extern "C" int bwts_inverse_device(bwts_ctx *ctx, const void *d_in, uint64_t n, void *d_out)
{
    return run_device(ctx, inverse_device_impl, d_in, n, d_out);
}

// ------------------------------------------------------------------------------------
// host-buffer entry points (the path of the CLIs: mk_bwts_sa.c:41-60, unbwts.c:29-89)
// ------------------------------------------------------------------------------------
// Caller memory (typically an mmap of the input file, and a fresh malloc for the output) is not pinned, so it cannot be
// a DMA target.  It moves through a ring of STAGE_SLOTS pinned buffers: while the DMA engine works on one slot the copy
// workers fill (or drain) the next, and a slot is reused once ITS event has fired -- nothing waits for the whole stream.
// Memory from bwts_host_alloc() is pinned already and is transferred in one piece.
#define STAGE_CHUNK ((size_t)8 << 20)

void CopyPool::part(int i)
{
    if (touching) {                   // workers 1 .. parts-1 share the buffer; the caller takes no part
        // MADV_POPULATE_WRITE maps the pages writable WITHOUT changing what they hold (a store of zeros, the first form, clobbered
        // the caller's buffer although the call could still fail, and -- in a batch -- an input that the output aliases).  Where the
        // kernel refuses it (before Linux 5.14, or a mapping it cannot populate) nothing is pre-faulted: the copy out pays the faults.
        const int nw = parts - 1;
        const uintptr_t b = ((uintptr_t)dst + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)dst + len) & ~(uintptr_t)4095;
        if (e <= b) return;
        const size_t pages = (e - b) >> 12, per = (pages + (size_t)nw - 1) / (size_t)nw;
        const size_t lo = per * (size_t)(i - 1) < pages ? per * (size_t)(i - 1) : pages;
        const size_t hi = lo + per < pages ? lo + per : pages;
        // (in pieces of 32 MiB: a worker that is told to stop gives up soon)
        for (size_t o = lo; o < hi && !touch_failed.load(std::memory_order_relaxed); o += 8192) {
            const size_t cnt = hi - o < 8192 ? hi - o : 8192;
            if (madvise((void *)(b + (o << 12)), cnt << 12, MADV_POPULATE_WRITE) != 0) touch_failed.store(true, std::memory_order_relaxed);
        }
        return;
    }
    // page-aligned cuts: two workers never fault on the same page of a fresh destination
    const size_t per = ((len / (size_t)parts) + 4095) & ~(size_t)4095;
    const size_t lo = per * (size_t)i < len ? per * (size_t)i : len;
    const size_t hi = i + 1 == parts ? len : (lo + per < len ? lo + per : len);
    if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
}

void CopyPool::start(int threads)
{
    parts = threads < 1 ? 1 : threads;
    for (int t = 1; t < parts; t++) {
        workers.emplace_back([this, t] {
            u64 seen = 0;
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_work.wait(lk, [&] { return stop || generation != seen; });
                    if (stop) return;
                    seen = generation;
                }
                part(t);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (--pending == 0) cv_done.notify_one();
                }
            }
        });
    }
}

void CopyPool::copy(void *d, const void *s_, size_t n)
{
    if (parts == 1 || n < ((size_t)1 << 20)) { memcpy(d, s_, n); return; }
    {
        std::lock_guard<std::mutex> lk(mu);
        dst = (char *)d; src = (const char *)s_; len = n;
        pending = parts - 1;
        generation++;
    }
    cv_work.notify_all();
    part(0);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
}

void CopyPool::touch_async(void *d, size_t n)
{
    if (parts == 1 || n < ((size_t)4 << 20)) return;
    {
        std::lock_guard<std::mutex> lk(mu);
        dst = (char *)d; src = nullptr; len = n;
        touching = true;
        touch_failed.store(false, std::memory_order_relaxed);
        pending = parts - 1;
        generation++;
    }
    cv_work.notify_all();
}

void CopyPool::wait()
{
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
    touching = false;
}

void CopyPool::shutdown()
{
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv_work.notify_all();
    for (std::thread &t : workers) t.join();
    workers.clear();
}

// Copies between HBM and pinned host memory by a kernel instead of the DMA engines: the SDMA path moved 29-30 GB/s from the
// device to the host on the MI355X boxes used here, stores issued by compute units reach the PCIe link's rate.
__global__ __launch_bounds__(256) void pcie_copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, u64 vecs,
                                                        u8 *__restrict__ dst_tail, const u8 *__restrict__ src_tail, u32 tail)
{
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (u64)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

static int copy_mode(const bwts_ctx *ctx, const char *name)      // 0 = DMA engine (hipMemcpyAsync), 1 = copy kernel
{
    const char *e = bwts_knob(ctx, name);
    if (e && !strcmp(e, "dma")) return 0;
    if (e && !strcmp(e, "kernel")) return 1;
    return -1;
}

static int pcie_copy(bwts_ctx *ctx, hipStream_t stream, void *dst, const void *src, size_t len, bool use_kernel, hipMemcpyKind kind)
{
    if (!use_kernel || (((uintptr_t)dst | (uintptr_t)src) & 15)) {
        HIPC(hipMemcpyAsync(dst, src, len, kind, stream));
        return BWTS_OK;
    }
    const u64 vecs = len / 16;
    const u32 tail = (u32)(len % 16);
    u64 blocks = (vecs + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    pcie_copy_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>((uint4 *)dst, (const uint4 *)src, vecs, (u8 *)dst + vecs * 16,
                                                                            (const u8 *)src + vecs * 16, tail);
    HIPC(hipGetLastError());
    return BWTS_OK;
}

static int ensure_staging(bwts_ctx *ctx, Stager &sg)
{
    const double t0 = wall_ms();
    const bool first = !sg.pool;
    if (!sg.stream) {
        if (&sg == &ctx->stg[0]) sg.stream = ctx->stream;
        else { HIPC(hipStreamCreateWithFlags(&sg.stream, hipStreamNonBlocking)); sg.own_stream = true; }
    }
    for (int i = 0; i < STAGE_SLOTS; i++) {
        if (!sg.pinned[i]) {
            void *p = nullptr;
            if (hipHostMalloc(&p, STAGE_CHUNK, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
            sg.pinned[i] = (char *)p;
            trace_alloc(ctx, "pinned +", "staging slot", p, STAGE_CHUNK);
        }
        if (!sg.slot_ev[i]) HIPC(hipEventCreateWithFlags(&sg.slot_ev[i], hipEventDisableTiming));
        if (!sg.slot_ev2[i]) HIPC(hipEventCreateWithFlags(&sg.slot_ev2[i], hipEventDisableTiming));
    }
    if (!sg.copy_stream) HIPC(hipStreamCreateWithFlags(&sg.copy_stream, hipStreamNonBlocking));
    if (!sg.pool) {
        int threads = 6;                                   // BWTS_COPY_THREADS: 1 = the calling thread alone
        if (const char *e = bwts_knob(ctx, "BWTS_COPY_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) threads = v; }
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)threads > hw) threads = (int)hw;
        sg.pool = new (std::nothrow) CopyPool();
        if (!sg.pool) return BWTS_E_NOMEM;
        sg.pool->start(threads);
    }
    if (first) ctx->host_ms[BWTS_H_STAGING_ALLOC] += wall_ms() - t0;
    return BWTS_OK;
}

static bool is_pinned_block(const bwts_ctx *ctx, const void *p, u64 n)
{
    for (const auto &b : ctx->host_blocks)
        if ((const char *)p >= b.first && (const char *)p + n <= b.first + b.second) return true;
    return false;
}

static int staged_h2d(bwts_ctx *ctx, Stager &sg, u8 *d_dst, const u8 *h_src, u64 n)
{
    const int mode = copy_mode(ctx, "BWTS_H2D");
    const bool by_kernel = mode == 1;
    BWTS_TRY(ensure_staging(ctx, sg));
    if (is_pinned_block(ctx, h_src, n)) {
        BWTS_TRY(pcie_copy(ctx, sg.stream, d_dst, h_src, n, by_kernel, hipMemcpyHostToDevice));
        HIPC(hipStreamSynchronize(sg.stream));
        return BWTS_OK;
    }
    u64 off = 0;
    for (u64 c = 0; off < n; c++) {
        const int slot = (int)(c % STAGE_SLOTS);
        const size_t len = n - off < STAGE_CHUNK ? (size_t)(n - off) : STAGE_CHUNK;
        if (c >= STAGE_SLOTS) HIPC(hipEventSynchronize(sg.slot_ev[slot]));       // the slot's previous DMA has read it
        sg.pool->copy(sg.pinned[slot], h_src + off, len);
        BWTS_TRY(pcie_copy(ctx, sg.stream, d_dst + off, sg.pinned[slot], len, by_kernel, hipMemcpyHostToDevice));
        HIPC(hipEventRecord(sg.slot_ev[slot], sg.stream));
        off += len;
    }
    HIPC(hipStreamSynchronize(sg.stream));
    return BWTS_OK;
}

// the result leaves in consecutive pieces: to h_dst (copy workers), or to the caller's sink straight from the staging slot
static int staged_d2h(bwts_ctx *ctx, Stager &sg, u8 *h_dst, const u8 *d_src, u64 n, bwts_sink_fn sink, void *user)
{
    const int mode = copy_mode(ctx, "BWTS_D2H");
    // default: the copy kernel for a single call (54 GB/s against the DMA engine's 29), the DMA engine inside a batch: there the
    // copy runs beside the next item's transform, whose kernels a copy kernel slows down by half (8 x 1 GiB: 18.0 GB/s with the
    // kernel, 24.4 GB/s with the engine -- the transform is the pipeline's slowest stage either way)
    const bool by_kernel = mode == 1 || (mode == -1 && &sg == &ctx->stg[0]);
    BWTS_TRY(ensure_staging(ctx, sg));
    if (!sink && is_pinned_block(ctx, h_dst, n)) {
        BWTS_TRY(pcie_copy(ctx, sg.stream, h_dst, d_src, n, by_kernel, hipMemcpyDeviceToHost));
        HIPC(hipStreamSynchronize(sg.stream));
        return BWTS_OK;
    }
    // a chunk's first part is moved by the copy kernel, the rest by the DMA engine on the second queue: the two paths
    // add up on the link (BWTS_D2H_SPLIT = percent moved by the kernel)
    const int kernel_pct = [ctx] { const char *e = bwts_knob(ctx, "BWTS_D2H_SPLIT"); int v = e ? atoi(e) : 100; return v < 0 ? 0 : v > 100 ? 100 : v; }();
    const u64 chunks = (n + STAGE_CHUNK - 1) / STAGE_CHUNK;
    auto issue = [&](u64 c) -> int {
        const u64 off = c * STAGE_CHUNK;
        const size_t len = n - off < STAGE_CHUNK ? (size_t)(n - off) : STAGE_CHUNK;
        const int slot = (int)(c % STAGE_SLOTS);
        size_t klen = by_kernel ? (len * (size_t)kernel_pct / 100) & ~(size_t)4095 : 0;
        if (by_kernel && kernel_pct == 100) klen = len;
        if (klen) BWTS_TRY(pcie_copy(ctx, sg.stream, sg.pinned[slot], d_src + off, klen, true, hipMemcpyDeviceToHost));
        HIPC(hipEventRecord(sg.slot_ev[slot], sg.stream));
        if (klen < len) HIPC(hipMemcpyAsync(sg.pinned[slot] + klen, d_src + off + klen, len - klen, hipMemcpyDeviceToHost, sg.copy_stream));
        HIPC(hipEventRecord(sg.slot_ev2[slot], sg.copy_stream));
        return BWTS_OK;
    };
    for (u64 c = 0; c < chunks && c < STAGE_SLOTS; c++) BWTS_TRY(issue(c));
    int rc = BWTS_OK;
    for (u64 c = 0; c < chunks; c++) {
        const u64 off = c * STAGE_CHUNK;
        const size_t len = n - off < STAGE_CHUNK ? (size_t)(n - off) : STAGE_CHUNK;
        const int slot = (int)(c % STAGE_SLOTS);
        if (hipEventSynchronize(sg.slot_ev[slot]) != hipSuccess || hipEventSynchronize(sg.slot_ev2[slot]) != hipSuccess) { rc = BWTS_E_HIP; break; }
        if (sink) { if (sink(user, (const uint8_t *)sg.pinned[slot], len) != 0) { rc = BWTS_E_SINK; break; } }
        else sg.pool->copy(h_dst + off, sg.pinned[slot], len);
        if (c + STAGE_SLOTS < chunks && (rc = issue(c + STAGE_SLOTS)) != BWTS_OK) break;
    }
    if (hipStreamSynchronize(sg.copy_stream) != hipSuccess && rc == BWTS_OK) rc = BWTS_E_HIP;
    if (hipStreamSynchronize(sg.stream) != hipSuccess && rc == BWTS_OK) rc = BWTS_E_HIP;
    return rc;
}


// device-side copies of the caller's input and output stay with the context (grown, never shrunk): d_io[0], d_io[1] inputs,
// d_io[2], d_io[3] outputs; the single calls use the first of each pair
static int ensure_io(bwts_ctx *ctx, u64 n, bool pairs)
{
    const double t0 = wall_ms();
    for (int i = 0; i < 4; i++) {
        if (!pairs && (i & 1)) continue;
        if (ctx->d_io_cap[i] >= n) continue;
        if (ctx->d_io[i]) { HIPC(hipStreamSynchronize(ctx->stream)); HIPC(ctx_free(ctx, ctx->d_io[i])); ctx->d_io[i] = nullptr; ctx->d_io_cap[i] = 0; }
        void *p = nullptr;
        const size_t cap = align_up((size_t)n, 1 << 20);
        static const char *const io_names[4] = {"device input 0", "device input 1", "device output 0", "device output 1"};
        if (ctx_malloc(ctx, &p, cap, io_names[i]) != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
        ctx->d_io[i] = (u8 *)p;
        ctx->d_io_cap[i] = cap;
    }
    ctx->host_ms[BWTS_H_IO_ALLOC] += wall_ms() - t0;
    return BWTS_OK;
}

static size_t arena_hint(device_impl_fn fn, u64 n)
{
    return n <= 0x100000000ull ? (fn == forward_device_impl ? forward_arena_bytes(n) : inverse_arena_bytes(n)) : 0;
}

static int run_host(bwts_ctx *ctx, device_impl_fn fn, const uint8_t *in, uint64_t n, uint8_t *out, bwts_sink_fn sink, void *user)
{
    if (!ctx || !in || (!out && !sink) || n == 0) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    BWTS_TRY(ensure_io(ctx, n, false));
    trace_alloc(ctx, "call    ", fn == forward_device_impl ? "forward: in" : "inverse: in", in, n);
    if (out) trace_alloc(ctx, "call    ", "out", out, n);
    Stager &sg = ctx->stg[0];
    // A context's first call allocates its arena, which can cost as long as the whole input copy where the driver clears what it
    // hands out: a helper thread does the hipMalloc -- and nothing else -- while this thread stages the input.  Rules (DESIGN.md
    // section 9: a GPU memory fault followed four arena re-reservations that an earlier form did wholly on the helper):
    //  * only a context that holds NO arena takes the helper (the one-shot CLI it was built for); a context that has to give up a
    //    smaller arena first does everything on this thread -- arena_release() drains the stream, then frees -- before staging starts;
    //  * the helper touches no context state: it allocates into a local, this thread installs the block after join().
    // BWTS_RESERVE_HELPER=1 (a test switch) sends EVERY growth through the helper, after the release on this thread.
    const size_t want = arena_hint(fn, n);
    std::thread reserve;
    void *fresh = nullptr;
    double reserve_ms = 0;
    if (want > ctx->arena_cap) {
        const char *force = bwts_knob(ctx, "BWTS_RESERVE_HELPER");
        const bool helper = !ctx->guard && (force ? force[0] == '1' : (ctx->arena == nullptr && want >= ((size_t)256 << 20)));
        if (helper) {
            BWTS_TRY(arena_release(ctx));
            const size_t bytes = align_up(want, 1 << 20);
            const int device = ctx->device;
            reserve = std::thread([device, bytes, &fresh, &reserve_ms] {
                const double t0 = wall_ms();
                void *p = nullptr;
                if (hipSetDevice(device) == hipSuccess && hipMalloc(&p, bytes) == hipSuccess) fresh = p;
                else (void)hipGetLastError();
                reserve_ms = wall_ms() - t0;
            });
        }
    }
    double t0 = wall_ms();
    const int h2d_rc = staged_h2d(ctx, sg, ctx->d_io[0], in, n);
    if (reserve.joinable()) {
        reserve.join();
        if (fresh) arena_install(ctx, fresh, align_up(want, 1 << 20), reserve_ms);     // (no block: the transform's own reservation reports it)
    }
    BWTS_TRY(h2d_rc);
    STAGE("host path: input on the device");
    const double h2d = wall_ms() - t0;
    // the caller's output buffer is usually fresh: its page faults are taken by the copy workers during the transform (contents
    // untouched: a call that fails leaves the buffer as it was, like the reference, which writes only after success: mk_bwts_sa.c:52-60)
    const bool touch = !sink && !is_pinned_block(ctx, out, n) && ensure_staging(ctx, sg) == BWTS_OK;
    if (touch) sg.pool->touch_async(out, n);
    const int rcd = run_device(ctx, fn, ctx->d_io[0], n, ctx->d_io[2]);
    if (touch) sg.pool->wait();
    BWTS_TRY(rcd);
    STAGE("host path: transform done");
    t0 = wall_ms();
    const int rc = staged_d2h(ctx, sg, out, ctx->d_io[2], n, sink, user);
    ctx->tm.d2h_ms = wall_ms() - t0;
    ctx->tm.h2d_ms = h2d;
    return rc;
}

// ------------------------------------------------------------------------------------
// batches of independent inputs on one context (BASELINE config 5's data flow per GPU): a three-stage pipeline
// ------------------------------------------------------------------------------------
// While item k is transformed (the calling thread, the context's stream), item k+1 is staged to the device by one helper thread
// (stg[1], its own queue and copy workers) and item k-1 is drained to the caller's buffer by another (stg[2]).  Two device buffers
// per direction; an item's input buffer is free again when its transform is done, its output buffer when it has been drained.
// Every item goes through exactly the code of the single calls, so the bytes are the same.
struct BatchState {
    std::mutex mu;
    std::condition_variable cv;
    int staged = -1, transformed = -1, drained = -1;     // highest item index that has passed each stage
    int error = BWTS_OK;
};

static int run_batch(bwts_ctx *ctx, device_impl_fn fn, int count, const uint8_t *const *ins, const uint64_t *ns, uint8_t *const *outs)
{
    if (!ctx || count < 0 || (count && (!ins || !ns || !outs))) return BWTS_E_ARG;
    u64 nmax = 0;
    for (int k = 0; k < count; k++) {
        if (!ins[k] || !outs[k] || ns[k] == 0) return BWTS_E_ARG;
        if (ns[k] > nmax) nmax = ns[k];
    }
    if (count == 0) return BWTS_OK;
    // An item's output may lie on its OWN input (in place, like `mk_bwts f f`: the input is on the device before a byte comes back) or on
    // an earlier item's.  It may not touch a LATER item's input: that item may not have been staged yet when this one is drained (single
    // calls in sequence would hand the later call the overwritten bytes; here it would be a race) -- refused.
    for (int k = 0; k < count; k++)
        for (int j = k + 1; j < count; j++)
            if (outs[k] < ins[j] + ns[j] && ins[j] < outs[k] + ns[k]) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    BWTS_TRY(ensure_io(ctx, nmax, true));
    BWTS_TRY(ensure_staging(ctx, ctx->stg[1]));
    BWTS_TRY(ensure_staging(ctx, ctx->stg[2]));
    BWTS_TRY(arena_reserve(ctx, arena_hint(fn, nmax) > ctx->arena_cap ? arena_hint(fn, nmax) : ctx->arena_cap));
    BatchState st;
    double busy[3] = {0, 0, 0};          // time the three stages spent working (BWTS_BATCH_TRACE=1 prints them)
    auto fail = [&st](int rc) { std::lock_guard<std::mutex> lk(st.mu); if (st.error == BWTS_OK) st.error = rc; st.cv.notify_all(); };
    const double t_begin = wall_ms();
    std::thread feeder([&] {
        if (hipSetDevice(ctx->device) != hipSuccess) { fail(BWTS_E_HIP); return; }
        for (int k = 0; k < count; k++) {
            {   // input buffer k % 2 is free once item k - 2 has been transformed
                std::unique_lock<std::mutex> lk(st.mu);
                st.cv.wait(lk, [&] { return st.error != BWTS_OK || st.transformed >= k - 2; });
                if (st.error != BWTS_OK) return;
            }
            const double t0 = wall_ms();
            const int rc = staged_h2d(ctx, ctx->stg[1], ctx->d_io[k & 1], ins[k], ns[k]);
            busy[0] += wall_ms() - t0;
            if (rc != BWTS_OK) { fail(rc); return; }
            { std::lock_guard<std::mutex> lk(st.mu); st.staged = k; }
            st.cv.notify_all();
        }
    });
    std::thread drainer([&] {
        if (hipSetDevice(ctx->device) != hipSuccess) { fail(BWTS_E_HIP); return; }
        for (int k = 0; k < count; k++) {
            // the caller's buffer is usually fresh: fault its pages in while the item is still being transformed (the pages' contents
            // stay as they are -- CopyPool::part -- so an output that lies on an input not yet staged does it no harm)
            const bool touch = !is_pinned_block(ctx, outs[k], ns[k]);
            if (touch) ctx->stg[2].pool->touch_async(outs[k], ns[k]);
            {
                std::unique_lock<std::mutex> lk(st.mu);
                st.cv.wait(lk, [&] { return st.error != BWTS_OK || st.transformed >= k; });
                if (touch) { lk.unlock(); ctx->stg[2].pool->wait(); lk.lock(); }
                if (st.error != BWTS_OK) return;
            }
            const double t0 = wall_ms();
            const int rc = staged_d2h(ctx, ctx->stg[2], outs[k], ctx->d_io[2 + (k & 1)], ns[k], nullptr, nullptr);
            busy[2] += wall_ms() - t0;
            if (rc != BWTS_OK) { fail(rc); return; }
            { std::lock_guard<std::mutex> lk(st.mu); st.drained = k; }
            st.cv.notify_all();
        }
    });
    for (int k = 0; k < count; k++) {
        {   // input k on the device, output buffer k % 2 drained of item k - 2
            std::unique_lock<std::mutex> lk(st.mu);
            st.cv.wait(lk, [&] { return st.error != BWTS_OK || (st.staged >= k && st.drained >= k - 2); });
            if (st.error != BWTS_OK) break;
        }
        const double t0 = wall_ms();
        const int rc = run_device(ctx, fn, ctx->d_io[k & 1], ns[k], ctx->d_io[2 + (k & 1)]);
        busy[1] += wall_ms() - t0;
        if (rc != BWTS_OK) { fail(rc); break; }
        { std::lock_guard<std::mutex> lk(st.mu); st.transformed = k; }
        st.cv.notify_all();
    }
    feeder.join();
    drainer.join();
    if (const char *e = bwts_knob(ctx, "BWTS_BATCH_TRACE"))
        if (e[0] == '1') fprintf(stderr, "[batch] %d items, wall %.1f ms; busy: copy in %.1f, transform %.1f, copy out %.1f ms\n", count, wall_ms() - t_begin, busy[0], busy[1], busy[2]);
    ctx->tm.h2d_ms = 0;
    ctx->tm.d2h_ms = wall_ms() - t_begin;        // (batch: wall time of the whole pipeline; total_ms is the last item's transform)
    return st.error;
}

extern "C" int bwts_forward_batch(bwts_ctx *ctx, int count, const uint8_t *const *ins, const uint64_t *ns, uint8_t *const *outs)
{
    return run_batch(ctx, forward_device_impl, count, ins, ns, outs);
}

extern "C" int bwts_inverse_batch(bwts_ctx *ctx, int count, const uint8_t *const *ins, const uint64_t *ns, uint8_t *const *outs)
{
    return run_batch(ctx, inverse_device_impl, count, ins, ns, outs);
}

extern "C" int bwts_forward(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out)
{
    return run_host(ctx, forward_device_impl, in, n, out, nullptr, nullptr);
}

extern "C" int bwts_inverse(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out)
{
    return run_host(ctx, inverse_device_impl, in, n, out, nullptr, nullptr);
}

extern "C" int bwts_forward_sink(bwts_ctx *ctx, const uint8_t *in, uint64_t n, bwts_sink_fn sink, void *user)
{
    if (!sink) return BWTS_E_ARG;
    return run_host(ctx, forward_device_impl, in, n, nullptr, sink, user);
}

extern "C" int bwts_inverse_sink(bwts_ctx *ctx, const uint8_t *in, uint64_t n, bwts_sink_fn sink, void *user)
{
    if (!sink) return BWTS_E_ARG;
    return run_host(ctx, inverse_device_impl, in, n, nullptr, sink, user);
}

extern "C" int bwts_host_alloc(bwts_ctx *ctx, uint64_t bytes, void **h_ptr)
{
    if (!ctx || !h_ptr) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
    ctx->host_blocks.emplace_back((char *)p, (size_t)bytes);
    *h_ptr = p;
    return BWTS_OK;
}

extern "C" int bwts_host_free(bwts_ctx *ctx, void *h_ptr)
{
    if (!ctx) return BWTS_E_ARG;
    if (!h_ptr) return BWTS_OK;
    for (size_t i = 0; i < ctx->host_blocks.size(); i++) {
        if (ctx->host_blocks[i].first == (char *)h_ptr) {
            ctx->host_blocks.erase(ctx->host_blocks.begin() + (long)i);
            HIPC(hipSetDevice(ctx->device));
            HIPC(hipHostFree(h_ptr));
            return BWTS_OK;
        }
    }
    return BWTS_E_ARG;
}

extern "C" int bwts_set_timing(bwts_ctx *ctx, int level)
{
    if (!ctx || level < 0 || level > 2) return BWTS_E_ARG;
    ctx->timing = level;
    return BWTS_OK;
}

// ------------------------------------------------------------------------------------
// introspection
// ------------------------------------------------------------------------------------
extern "C" int bwts_last_timings(bwts_ctx *ctx, bwts_timings *t)
{
    if (!ctx || !t) return BWTS_E_ARG;
    *t = ctx->tm;
    for (int i = 0; i < BWTS_H_COUNT; i++) t->host_ms[i] = ctx->host_ms[i];
    return BWTS_OK;
}

extern "C" const char *bwts_kernel_class_name(int k)
{
    static const char *names[BWTS_K_COUNT] = {"histogram", "keybuild", "radix_hist", "radix_scan", "radix_scatter", "rerank",
                                              "lyndon", "emit", "lf_build", "walk", "listrank", "walk_emit", "other", "radix_scatter_main", "round"};
    return (k >= 0 && k < BWTS_K_COUNT) ? names[k] : "?";
}

extern "C" const char *bwts_host_cost_name(int h)
{
    static const char *names[BWTS_H_COUNT] = {"init", "module_load", "io_alloc", "staging_alloc", "arena_alloc"};
    return (h >= 0 && h < BWTS_H_COUNT) ? names[h] : "?";
}

extern "C" const char *bwts_strerror(int code)
{
    switch (code) {
    case BWTS_OK: return "ok";
    case BWTS_E_ARG: return "invalid argument (null pointer or empty input)";
    case BWTS_E_NODEVICE: return "no usable HIP device";
    case BWTS_E_NOMEM: return "out of device or pinned host memory";
    case BWTS_E_HIP: return "HIP runtime error";
    case BWTS_E_RANGE: return "input length beyond the engine's index range";
    case BWTS_E_INTERNAL: return "internal invariant violated";
    case BWTS_E_SINK: return "the caller's output sink reported an error";
    default: return "unknown error";
    }
}

extern "C" int bwts_last_hip_error(bwts_ctx *ctx) { return ctx ? ctx->last_hip : 0; }

// ------------------------------------------------------------------------------------
// harness utilities
// ------------------------------------------------------------------------------------
extern "C" int bwts_generate_device(bwts_ctx *ctx, int kind, uint64_t seed, uint64_t n, void *d_out)
{
    if (!ctx || !d_out) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    return generate_device_impl(ctx, kind, seed, n, (u8 *)d_out);
}

extern "C" int bwts_device_alloc(bwts_ctx *ctx, uint64_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
    return BWTS_OK;
}

extern "C" int bwts_device_free(bwts_ctx *ctx, void *d_ptr)
{
    if (!ctx) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    HIPC(hipFree(d_ptr));
    return BWTS_OK;
}

extern "C" int bwts_copy_to_device(bwts_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !h_src))) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    HIPC(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return BWTS_OK;
}

extern "C" int bwts_copy_to_host(bwts_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!h_dst || !d_src))) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    HIPC(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return BWTS_OK;
}

extern "C" int bwts_device_equal(bwts_ctx *ctx, const void *d_a, const void *d_b, uint64_t bytes, int *equal)
{
    if (!ctx || !equal) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    return device_equal_impl(ctx, (const u8 *)d_a, (const u8 *)d_b, bytes, equal);
}

// ------------------------------------------------------------------------------------
// unit-test hooks
// ------------------------------------------------------------------------------------
extern "C" int bwts_debug_sort_pairs(bwts_ctx *ctx, uint64_t *h_keys, uint32_t *h_vals, uint64_t m, int key_bits)
{
    if (!ctx || !h_keys || !h_vals) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    spans_reset(ctx);
    const size_t need = 2 * align_up(m * 8, 256) + 2 * align_up(m * 4, 256) + radix_tile_hist_bytes(m) + scan_temp_bytes(m) + 4096;
    BWTS_TRY(arena_reserve(ctx, need));
    SortPlan plan;
    plan.keys[0] = arena_array<u64>(ctx, m); plan.keys[1] = arena_array<u64>(ctx, m);
    plan.vals[0] = arena_array<u32>(ctx, m); plan.vals[1] = arena_array<u32>(ctx, m);
    plan.tile_hist = (u32 *)arena_alloc(ctx, radix_tile_hist_bytes(m));
    plan.scan_temp = arena_alloc(ctx, scan_temp_bytes(m));
    if (!plan.keys[0] || !plan.keys[1] || !plan.vals[0] || !plan.vals[1] || !plan.tile_hist || !plan.scan_temp) return BWTS_E_NOMEM;
    HIPC(hipMemcpyAsync(plan.keys[0], h_keys, m * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(plan.vals[0], h_vals, m * 4, hipMemcpyHostToDevice, ctx->stream));
    int res = 0;
    BWTS_TRY(radix_sort_pairs(ctx, plan, m, key_bits, &res));
    HIPC(hipMemcpyAsync(h_keys, plan.keys[res], m * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(h_vals, plan.vals[res], m * 4, hipMemcpyDeviceToHost, ctx->stream));
    BWTS_TRY(spans_resolve(ctx));
    return BWTS_OK;
}

static int upload_text(bwts_ctx *ctx, const uint8_t *in, uint64_t n, u8 **d_T)
{
    void *p = nullptr;
    if (hipMalloc(&p, n) != hipSuccess) { (void)hipGetLastError(); return BWTS_E_NOMEM; }
    if (hipMemcpyAsync(p, in, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipFree(p); return BWTS_E_HIP; }
    *d_T = (u8 *)p;
    return BWTS_OK;
}

extern "C" int bwts_debug_suffix_array(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint32_t *h_sa)
{
    if (!ctx || !in || !h_sa || n == 0) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    spans_reset(ctx);
    u8 *d_T = nullptr;
    BWTS_TRY(upload_text(ctx, in, n, &d_T));
    int rc = arena_reserve(ctx, forward_arena_bytes(n));
    u32 *sa = nullptr, *rank = nullptr, rounds = 0;
    if (rc == BWTS_OK) rc = suffix_sort_device(ctx, d_T, n, &sa, &rank, &rounds);
    if (rc == BWTS_OK && hipMemcpyAsync(h_sa, sa, n * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = BWTS_E_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == BWTS_OK) rc = BWTS_E_HIP;
    (void)hipFree(d_T);
    return rc;
}

extern "C" int bwts_debug_lyndon(bwts_ctx *ctx, const uint8_t *in, uint64_t n, uint64_t *h_starts, uint64_t cap, uint64_t *count)
{
    if (!ctx || !in || !h_starts || !count || n == 0) return BWTS_E_ARG;
    HIPC(hipSetDevice(ctx->device));
    spans_reset(ctx);
    u8 *d_T = nullptr;
    BWTS_TRY(upload_text(ctx, in, n, &d_T));
    int rc = arena_reserve(ctx, forward_arena_bytes(n));
    u32 *fstart = nullptr, rounds = 0;
    u64 k = 0;
    if (rc == BWTS_OK) rc = lyndon_factors_device(ctx, d_T, n, &fstart, &k, &rounds);
    if (rc == BWTS_OK) {
        const u64 take = k < cap ? k : cap;
        u32 *tmp = (u32 *)malloc((size_t)(take ? take : 1) * 4);
        if (!tmp) rc = BWTS_E_NOMEM;
        else {
            // on the context's stream: it is non-blocking, so the null stream would not wait for the factor list's last copy
            if (hipMemcpyAsync(tmp, fstart, take * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) rc = BWTS_E_HIP;
            for (u64 i = 0; i < take; i++) h_starts[i] = tmp[i];
            free(tmp);
        }
        *count = k;
    }
    (void)hipFree(d_T);
    return rc;
}
