// radix_digits.hip -- ONE stable LSD radix pass over (u32 key, u32 value, u16 extra) streams -- the layout of the engine's packed
// round-0 passes: 10 bytes in, 10 bytes out per element -- with DB-bit digits, DB = 8 (the product), 10 and 11, in the product's
// kernel structure (per-tile histogram kernel, column scan, scatter kernel with wave-ballot ranking, LDS staging in sorted order,
// slot-consecutive write-out, XCD-contiguous tile ranges), so that the question "do 40-bit keys sort faster in 4 passes of 10 bits
// than in 5 of 8?" gets a measured answer per pass instead of an estimate (review of round 2, item 4).
//   hipcc -O3 --offload-arch=gfx950 radix_digits.hip -o radix_digits && ./radix_digits [log2n]
// Prints per configuration: histogram ms, scan ms, scatter ms, total per pass, and the time of a whole 40-bit sort at that rate
// (5 x 8 bits, 4 x 10 bits, 4 x 11 bits cover 40 / 40 / 44 bits).  Checks every pass against a host reference at 2^20.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32; typedef uint64_t u64; typedef uint16_t u16; typedef uint8_t u8;

__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
template <int DB> __device__ __forceinline__ u64 match_digit(u32 digit, bool valid)
{
    const u64 vm = __ballot(valid);
    u32 plo = (u32)vm, phi = (u32)(vm >> 32);
#pragma unroll
    for (int b = 0; b < DB; b++) {
        const u32 t = (u32)__builtin_amdgcn_sbfe((int)digit, (unsigned)b, 1u);
        const u64 m = __ballot((int)t < 0);
        plo &= ~((u32)m ^ t);
        phi &= ~((u32)(m >> 32) ^ t);
    }
    return ((u64)phi << 32) | plo;
}
__device__ __forceinline__ u32 wave_incl_scan(u32 v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_up((int)v, d, 64); if (lane >= d) v += o; }
    return v;
}

template <int TH, int IT, int DB>
__global__ __launch_bounds__(TH) void hist_kernel(const u32 *__restrict__ keys, u64 m, int shift, u32 *__restrict__ tile_hist)
{
    constexpr int NB = 1 << DB, WV = TH / 64, TILE = TH * IT;
    __shared__ u32 bins[WV][NB];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < WV * NB; i += TH) ((u32 *)bins)[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * TILE;
    u32 k[IT];
#pragma unroll
    for (int j = 0; j < IT; j++) { const u64 i = base + (u64)j * TH + tid; k[j] = i < m ? keys[i] : 0u; }
#pragma unroll
    for (int j = 0; j < IT; j++) { const u64 i = base + (u64)j * TH + tid; if (i < m) atomicAdd(&bins[w][(k[j] >> shift) & (NB - 1)], 1u); }
    __syncthreads();
    for (int d = tid; d < NB; d += TH) { u32 s = 0; for (int ww = 0; ww < WV; ww++) s += bins[ww][d]; tile_hist[(u64)blockIdx.x * NB + d] = s; }
}

// [tile][digit] counts -> exclusive offsets in digit-major order (simple two-level form: per digit a serial walk over the tiles in
// chunks; its cost is reported separately and is small next to the sweeps)
template <int DB>
__global__ __launch_bounds__(256) void colsum_kernel(const u32 *__restrict__ tile_hist, u64 tiles, u32 *__restrict__ digit_tot)
{
    constexpr int NB = 1 << DB;
    const u32 d = blockIdx.x;                 // one workgroup per digit
    __shared__ u32 sm[4];
    u32 s = 0;
    for (u64 t = threadIdx.x; t < tiles; t += 256) s += tile_hist[t * NB + d];
    s = wave_incl_scan(s);
    if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) digit_tot[d] = sm[0] + sm[1] + sm[2] + sm[3];
}
template <int DB>
__global__ __launch_bounds__(256) void colapply_kernel(u32 *__restrict__ tile_hist, u64 tiles, const u32 *__restrict__ digit_tot)
{
    constexpr int NB = 1 << DB;
    const u32 d = blockIdx.x;
    __shared__ u32 sm[4];
    __shared__ u32 base_s;
    if (threadIdx.x == 0) { u32 b = 0; for (u32 q = 0; q < d; q++) b += digit_tot[q]; base_s = b; }
    __syncthreads();
    u32 run = base_s;
    for (u64 t0 = 0; t0 < tiles; t0 += 256) {
        const u64 t = t0 + threadIdx.x;
        const u32 v = t < tiles ? tile_hist[t * NB + d] : 0u;
        const u32 inc = wave_incl_scan(v);
        if ((threadIdx.x & 63) == 63) sm[threadIdx.x >> 6] = inc;
        __syncthreads();
        u32 wp = 0, tot = 0;
        for (int w = 0; w < 4; w++) { if (w < (int)(threadIdx.x >> 6)) wp += sm[w]; tot += sm[w]; }
        if (t < tiles) tile_hist[t * NB + d] = run + wp + inc - v;
        run += tot;
        __syncthreads();
    }
}

// scatter: the product kernel's structure with DB-bit digits.  LDS: staged pair (8 B per element, the u16 rides in a second
// trip like the product's), per-wave 16-bit digit counters, digit base tables, per-slot digit table (u8 or u16).
template <int TH, int IT, int DB, int MINW>
__global__ __launch_bounds__(TH, MINW) void scatter_kernel(const u32 *__restrict__ kin, const u32 *__restrict__ vin, const u16 *__restrict__ cin,
                                                           u32 *__restrict__ kout, u32 *__restrict__ vout, u16 *__restrict__ cout,
                                                           const u32 *__restrict__ tile_off, u64 m, int shift)
{
    constexpr int NB = 1 << DB, WV = TH / 64, TILE = TH * IT;
    typedef typename std::conditional<(DB > 8), u16, u8>::type dig_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *stage = (u64 *)smem;                                  // TILE x 8 B
    u32 *dbase = (u32 *)(stage + TILE);                        // NB
    u32 *gbase = dbase + NB;                                   // NB
    u32 *scan_sm = gbase + NB;                                 // 16
    u16 (*whist)[NB] = (u16 (*)[NB])(scan_sm + 16);            // WV x NB
    dig_t *sdig = (dig_t *)(whist + WV);                       // TILE
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u64 tiles = (m + TILE - 1) / TILE;
    const u64 per = (tiles + 7) / 8;
    const u64 tile = (u64)(blockIdx.x % 8) * per + (u64)(blockIdx.x / 8);        // XCD-contiguous tile ranges
    const u64 tile_base = tile * TILE;
    if (tile >= tiles || tile_base >= m) return;
    const u64 wave_base = tile_base + (u64)w * (64 * IT);
    const u64 remain = m - tile_base;
    const u32 tile_count = remain < TILE ? (u32)remain : (u32)TILE;
    for (int i = tid; i < WV * NB / 2; i += TH) ((u32 *)whist)[i] = 0;
    u32 key[IT];
    u32 posp[IT / 2];
#pragma unroll
    for (int j = 0; j < IT; j++) { const u64 i = wave_base + (u64)j * 64 + lane; key[j] = i < m ? kin[i] : ~0u; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < m;
        const u32 d = (key[j] >> shift) & (NB - 1);
        const u64 peers = match_digit<DB>(d, valid);
        const u32 before = (u32)__popcll(peers & lanemask_lt());
        const u32 cnt = (u32)__popcll(peers);
        const u32 prev = whist[w][d];
        if (j & 1) posp[j >> 1] |= (prev + before) << 16; else posp[j >> 1] = prev + before;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && before == 0) whist[w][d] = (u16)(prev + cnt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __syncthreads();
    {
        // digits [tid * NB / TH, ...) per thread: exclusive scan over (digit, wave)
        constexpr int DPT = NB / TH > 0 ? NB / TH : 1;
        u32 run = 0;
        u32 loc[DPT];
        const bool own = tid * DPT < NB;
#pragma unroll
        for (int q = 0; q < DPT; q++) {
            loc[q] = run;
            if (own) {
                const int d = tid * DPT + q;
                for (int ww = 0; ww < WV; ww++) { const u32 c = whist[ww][d]; whist[ww][d] = (u16)(run - loc[q]); run += c; }
            }
        }
        // block exclusive scan of run
        const u32 inc = wave_incl_scan(run);
        if (lane == 63) scan_sm[w] = inc;
        __syncthreads();
        u32 wp = 0;
        for (int ww = 0; ww < WV; ww++) if (ww < w) wp += scan_sm[ww];
        const u32 exc = wp + inc - run;
        if (own) {
#pragma unroll
            for (int q = 0; q < DPT; q++) {
                const int d = tid * DPT + q;
                dbase[d] = exc + loc[q];
                gbase[d] = tile_off[tile * NB + d] - (exc + loc[q]);
            }
        }
    }
    __syncthreads();
#define POS_GET(j) (((j) & 1) ? posp[(j) >> 1] >> 16 : posp[(j) >> 1] & 0xffffu)
#define POS_SET(j, v) do { if ((j) & 1) posp[(j) >> 1] = (posp[(j) >> 1] & 0xffffu) | ((v) << 16); else posp[(j) >> 1] = (posp[(j) >> 1] & 0xffff0000u) | (v); } while (0)
    u32 *stage32 = (u32 *)stage;
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < m;
        const u32 d = (key[j] >> shift) & (NB - 1);
        const u32 p = POS_GET(j) + dbase[d] + whist[w][d];
        POS_SET(j, p);
        if (valid) { stage32[p] = key[j]; sdig[p] = (dig_t)d; }
    }
    u32 val[IT], ext[IT];
#pragma unroll
    for (int j = 0; j < IT; j++) { const u64 i = wave_base + (u64)j * 64 + lane; val[j] = i < m ? vin[i] : 0u; }
#pragma unroll
    for (int j = 0; j < IT; j++) { const u64 i = wave_base + (u64)j * 64 + lane; ext[j] = i < m ? (u32)cin[i] : 0u; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const u32 s = (u32)j * TH + tid;
        if (s < tile_count) kout[gbase[sdig[s]] + s] = stage32[s];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const bool valid = wave_base + (u64)j * 64 + lane < m;
        if (valid) ((uint2 *)stage)[POS_GET(j)] = make_uint2(val[j], ext[j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IT; j++) {
        const u32 s = (u32)j * TH + tid;
        if (s < tile_count) {
            const uint2 e = ((const uint2 *)stage)[s];
            const u32 dst = gbase[sdig[s]] + s;
            vout[dst] = e.x;
            cout[dst] = (u16)e.y;
        }
    }
#undef POS_GET
#undef POS_SET
}

__global__ void fill_kernel(u32 *k, u32 *v, u16 *c, u64 m, u64 seed)
{
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < m; i += gridDim.x * 256ull) {
        u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        k[i] = (u32)z; v[i] = (u32)i; c[i] = (u16)(z >> 40);
    }
}

template <int TH, int IT, int DB, int MINW>
static double run_config(const char *name, int log2n, u32 *k[2], u32 *v[2], u16 *c[2], u32 *tile_hist, u32 *digit_tot, int passes_for_40, bool verify)
{
    constexpr int NB = 1 << DB, TILE = TH * IT;
    const u64 m = 1ull << log2n, tiles = (m + TILE - 1) / TILE;
    const size_t lds = (size_t)TILE * 8 + 2 * NB * 4 + 64 + (size_t)(TH / 64) * NB * 2 + (size_t)TILE * (DB > 8 ? 2 : 1);
    CK(hipFuncSetAttribute((const void *)scatter_kernel<TH, IT, DB, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scatter_kernel<TH, IT, DB, MINW>, TH, lds));
    hipEvent_t ev[4]; for (auto &x : ev) CK(hipEventCreate(&x));
    float th = 0, ts = 0, tc = 0;
    const int shift = 3;                               // any digit position: the keys are uniform
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(ev[0]));
        hist_kernel<TH, IT, DB><<<dim3((unsigned)tiles), dim3(TH)>>>(k[0], m, shift, tile_hist);
        CK(hipEventRecord(ev[1]));
        colsum_kernel<DB><<<dim3(NB), dim3(256)>>>(tile_hist, tiles, digit_tot);
        colapply_kernel<DB><<<dim3(NB), dim3(256)>>>(tile_hist, tiles, digit_tot);
        CK(hipEventRecord(ev[2]));
        scatter_kernel<TH, IT, DB, MINW><<<dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(TH), lds>>>(k[0], v[0], c[0], k[1], v[1], c[1], tile_hist, m, shift);
        CK(hipEventRecord(ev[3])); CK(hipEventSynchronize(ev[3])); CK(hipGetLastError());
        CK(hipEventElapsedTime(&th, ev[0], ev[1])); CK(hipEventElapsedTime(&ts, ev[1], ev[2])); CK(hipEventElapsedTime(&tc, ev[2], ev[3]));
    }
    bool ok = true;
    if (verify) {
        std::vector<u32> hk(m), hv(m), ok_(m), ov(m); std::vector<u16> hc(m), oc(m);
        CK(hipMemcpy(hk.data(), k[0], m * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv.data(), v[0], m * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc.data(), c[0], m * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ok_.data(), k[1], m * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ov.data(), v[1], m * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(oc.data(), c[1], m * 2, hipMemcpyDeviceToHost));
        std::vector<u32> idx(m); for (u64 i = 0; i < m; i++) idx[i] = (u32)i;
        std::stable_sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return ((hk[a] >> shift) & (NB - 1)) < ((hk[b] >> shift) & (NB - 1)); });
        for (u64 i = 0; i < m && ok; i++) ok = ok_[i] == hk[idx[i]] && ov[i] == hv[idx[i]] && oc[i] == hc[idx[i]];
    }
    const double pass = th + ts + tc;
    printf("%-34s n=2^%d  LDS %3zu KB  workgroups/CU %d  hist %6.3f  scan %6.3f  scatter %6.3f  pass %6.3f ms = %5.2f TB/s on 20 B  | 40-bit sort: %d passes = %6.2f ms%s\n",
           name, log2n, lds >> 10, occ, th, ts, tc, pass, 20.0 * m / (tc * 1e9), passes_for_40, passes_for_40 * pass, verify ? (ok ? "  [exact]" : "  [MISMATCH]") : "");
    fflush(stdout);
    return pass;
}

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const u64 m = 1ull << log2n;
    u32 *k[2], *v[2]; u16 *c[2]; u32 *tile_hist, *digit_tot;
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&k[i], m * 4)); CK(hipMalloc(&v[i], m * 4)); CK(hipMalloc(&c[i], m * 2)); }
    CK(hipMalloc(&tile_hist, ((m + 2047) / 2048 + 8) * 2048 * 4)); CK(hipMalloc(&digit_tot, 2048 * 4));
    fill_kernel<<<4096, 256>>>(k[0], v[0], c[0], m, 12345);
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 2; round++) {
        const bool verify = round == 0;
        const int L = verify ? 20 : log2n;
        if (verify && log2n < 20) continue;
        run_config<512, 16, 8, 4>("8-bit digits, 512 x 16 (product)", L, k, v, c, tile_hist, digit_tot, 5, verify);
        run_config<256, 16, 8, 4>("8-bit digits, 256 x 16", L, k, v, c, tile_hist, digit_tot, 5, verify);
        run_config<512, 16, 10, 2>("10-bit digits, 512 x 16", L, k, v, c, tile_hist, digit_tot, 4, verify);
        run_config<1024, 8, 10, 4>("10-bit digits, 1024 x 8", L, k, v, c, tile_hist, digit_tot, 4, verify);
        run_config<512, 8, 10, 4>("10-bit digits, 512 x 8", L, k, v, c, tile_hist, digit_tot, 4, verify);
        run_config<512, 16, 11, 2>("11-bit digits, 512 x 16", L, k, v, c, tile_hist, digit_tot, 4, verify);
    }
    return 0;
}
