// walk_steps.hip -- from a bare pointer chase towards the inverse walk's step, one ingredient at a time:
//   bit0  dense per-wave index log (one contiguous store per wave and step)
//   bit1  16-byte symbol store every 16 steps into a per-lane slot
//   bit2  8-step binary search in an LDS table per step
//   bit3  segment ends: a lane restarts from a fresh index every ~512 steps (data-dependent exit + 4 small stores)
//   bit4  index log in lane-fixed form: a lane keeps four indices and stores them as one 16-byte word every fourth step -- the wave's
//         store is one contiguous KB per four steps instead of 256 B every step
//   bit5  symbol store in wave-coalesced form: every 16 steps all lanes store their 16 symbols side by side (one contiguous KB per wave)
//         instead of 16 bytes each into slots of their own
//   hipcc -O3 --offload-arch=gfx950 walk_steps.hip -o walk_steps && ./walk_steps
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u32 mix(u64 z) { z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull; z ^= z >> 33; return (u32)z; }
__global__ void fill(u32 *t, u64 n, u32 mask) { for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) t[i] = mix(i) & mask; }

template <int F> __global__ __launch_bounds__(256) void chase(const u32 *__restrict__ t, u32 mask, int steps, u32 *out, u32 *log, uint4 *seg, u32 *nodes)
{
    __shared__ u64 tab[257];
    for (int i = threadIdx.x; i < 257; i += 256) tab[i] = (u64)i << 22;
    __syncthreads();
    const u64 gid = blockIdx.x * 256ull + threadIdx.x;
    const u64 wave = gid >> 6;
    const int lane = threadIdx.x & 63;
    u32 x = mix(gid + 12345) & mask;
    u32 sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, len = 0, restarts = 0;
    u64 lcur = wave * (u64)steps * 64;
    u32 l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    u32 big[16];
#pragma unroll
    for (int q = 0; q < 16; q++) big[q] = 0;
    for (int s = 0; s < steps; s++) {
        const u32 y = t[x];
        if (F & 16) {
            const int q = s & 3;
            l0 = q == 0 ? x : l0; l1 = q == 1 ? x : l1; l2 = q == 2 ? x : l2; l3 = q == 3 ? x : l3;
            if (q == 3) ((uint4 *)log)[(wave * (u64)(steps / 4) + (u64)(s >> 2)) * 64 + lane] = make_uint4(l0, l1, l2, l3);
        }
        if (F & 1) {
            const u64 act = __ballot(true);
            log[lcur + __popcll(act & ((1ull << lane) - 1))] = x;
            lcur += __popcll(act);
        }
        u32 sym = y & 255u;
        if (F & 4) {
            u32 lo = 0, hi = 255;
#pragma unroll
            for (int it = 0; it < 8; it++) { const u32 mid = (lo + hi + 1) >> 1; if (tab[mid] <= (u64)y) lo = mid; else hi = mid - 1; }
            sym = lo;
        }
        if (F & 2) {
            const u32 sh = sym << (8 * (len & 3u));
            const u32 w = (len >> 2) & 3u;
            sb0 |= w == 0 ? sh : 0u; sb1 |= w == 1 ? sh : 0u; sb2 |= w == 2 ? sh : 0u; sb3 |= w == 3 ? sh : 0u;
            if ((len & 15u) == 15u) { seg[gid * (steps / 16 + 1) + (len >> 4)] = make_uint4(sb0, sb1, sb2, sb3); sb0 = sb1 = sb2 = sb3 = 0; }
        } else if (F & 64) {       // per-lane slots as in bit1, but 64 symbols gathered in 16 registers and stored as four 16-byte words at once
            const u32 sh = sym << (8 * (len & 3u));
            const u32 w = (len >> 2) & 15u;
#pragma unroll
            for (int q = 0; q < 16; q++) big[q] |= w == (u32)q ? sh : 0u;
            if ((len & 63u) == 63u) {
                uint4 *d = seg + gid * (steps / 16 + 4) + (len >> 6) * 4;
#pragma unroll
                for (int q = 0; q < 4; q++) d[q] = make_uint4(big[4 * q], big[4 * q + 1], big[4 * q + 2], big[4 * q + 3]);
#pragma unroll
                for (int q = 0; q < 16; q++) big[q] = 0;
            }
        } else if (F & 32) {
            const u32 sh = sym << (8 * (s & 3));
            const u32 w = (s >> 2) & 3;
            sb0 |= w == 0 ? sh : 0u; sb1 |= w == 1 ? sh : 0u; sb2 |= w == 2 ? sh : 0u; sb3 |= w == 3 ? sh : 0u;
            if ((s & 15) == 15) { seg[(wave * (u64)(steps / 16 + 1) + (u64)(s >> 4)) * 64 + lane] = make_uint4(sb0, sb1, sb2, sb3); sb0 = sb1 = sb2 = sb3 = 0; }
        } else sb0 += sym;
        len++;
        x = y;
        if (F & 8) {
            if ((x & 511u) == 0) {      // "splitter": close the node, start elsewhere
                const u64 nd = gid * 8 + (restarts & 7);
                nodes[nd * 4] = x; nodes[nd * 4 + 1] = len; nodes[nd * 4 + 2] = sb0; nodes[nd * 4 + 3] = restarts;
                restarts++;
                x = mix(gid * 977 + restarts) & mask; 
            }
        }
    }
    if (F & 64) { for (int q = 0; q < 16; q++) sb0 += big[q]; }
    if (x == 0xffffffffu || sb0 == 0x12345) out[0] = x;
}
template <int F> void run(const u32 *t, u32 mask, u32 *out, u32 *log, uint4 *seg, u32 *nodes, int blocks, int steps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    chase<F><<<blocks, 256>>>(t, mask, 16, out, log, seg, nodes);
    CK(hipEventRecord(a));
    chase<F><<<blocks, 256>>>(t, mask, steps, out, log, seg, nodes);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("features %2d : %7.2f ms  %6.1f G steps/s\n", F, ms, (double)blocks * 256 * steps / ms * 1e-6);
}
__global__ void gen_text(uint8_t *b, u32 *iota, u64 n, int skew) { for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) { u32 r = mix(i * 3 + 1); float u = (r >> 8) * (1.0f / 16777216.0f); b[i] = skew ? (uint8_t)(255.99f * u * u * u * u) : (uint8_t)(r >> 24); iota[i] = (u32)i; } }
__global__ void invert(const u32 *src, u32 *lf, u64 n) { for (u64 j = blockIdx.x * 256ull + threadIdx.x; j < n; j += gridDim.x * 256ull) lf[src[j]] = (u32)j; }
// LF of a byte text: stable sort of (byte, index); sorted slot j holds index i  ->  LF[i] = j
static void build_lf(u32 *t, u64 n, int skew)
{
    uint8_t *b, *b2; u32 *io, *so; CK(hipMalloc(&b, n)); CK(hipMalloc(&b2, n)); CK(hipMalloc(&io, n * 4)); CK(hipMalloc(&so, n * 4));
    gen_text<<<8192, 256>>>(b, io, n, skew);
    size_t tb = 0; void *tmp = nullptr;
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, b, b2, io, so, (int)n));   // n < 2^31
    CK(hipMalloc(&tmp, tb));
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, b, b2, io, so, (int)n));
    invert<<<8192, 256>>>(so, t, n);
    CK(hipDeviceSynchronize());
    CK(hipFree(b)); CK(hipFree(b2)); CK(hipFree(io)); CK(hipFree(so)); CK(hipFree(tmp));
}
int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30;
    const u64 n = 1ull << log2n; const u32 mask = (u32)(n - 1);
    const int blocks = 2048, steps = 512;
    const u64 lanes = (u64)blocks * 256;
    u32 *out, *t, *log, *nodes; uint4 *seg;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&t, n * 4));
    CK(hipMalloc(&log, lanes * steps * 4)); CK(hipMalloc(&seg, lanes * (steps / 16 + 4) * 16)); CK(hipMalloc(&nodes, lanes * 8 * 16));
  for (int kind = 0; kind < 3; kind++) {
    printf("table kind %d (0 random function, 1 LF of uniform bytes, 2 LF of skewed bytes)\n", kind);
    if (kind == 0) { fill<<<8192, 256>>>(t, n, mask); CK(hipDeviceSynchronize()); } else build_lf(t, n, kind == 2);
    run<0>(t, mask, out, log, seg, nodes, blocks, steps);
    run<1>(t, mask, out, log, seg, nodes, blocks, steps);
    run<2>(t, mask, out, log, seg, nodes, blocks, steps);
    run<4>(t, mask, out, log, seg, nodes, blocks, steps);
    run<8>(t, mask, out, log, seg, nodes, blocks, steps);
    run<3>(t, mask, out, log, seg, nodes, blocks, steps);
    run<6>(t, mask, out, log, seg, nodes, blocks, steps);
    run<7>(t, mask, out, log, seg, nodes, blocks, steps);
    run<15>(t, mask, out, log, seg, nodes, blocks, steps);
    run<16>(t, mask, out, log, seg, nodes, blocks, steps);
    run<32>(t, mask, out, log, seg, nodes, blocks, steps);
    run<48>(t, mask, out, log, seg, nodes, blocks, steps);
    run<33>(t, mask, out, log, seg, nodes, blocks, steps);
    run<60>(t, mask, out, log, seg, nodes, blocks, steps);
    run<45>(t, mask, out, log, seg, nodes, blocks, steps);
    run<64>(t, mask, out, log, seg, nodes, blocks, steps);
    run<65>(t, mask, out, log, seg, nodes, blocks, steps);
    run<77>(t, mask, out, log, seg, nodes, blocks, steps);
  }
    return 0;
}
