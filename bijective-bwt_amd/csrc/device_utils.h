// device_utils.h -- wave64 / workgroup building blocks for gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;
typedef uint8_t  u8;

#define WAVE 64

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

__device__ __forceinline__ u64 lanemask_lt()
{
    return (1ull << lane_id()) - 1ull;
}

struct OpAdd { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct OpMax { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct OpMin { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a < b ? a : b; } };

__device__ __forceinline__ u32 shfl_up_t(u32 v, int d) { return (u32)__shfl_up((int)v, d, 64); }
__device__ __forceinline__ u64 shfl_up_t(u64 v, int d)
{
    u32 lo = (u32)__shfl_up((int)(u32)v, d, 64);
    u32 hi = (u32)__shfl_up((int)(u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_down_t(u64 v, int d)
{
    u32 lo = (u32)__shfl_down((int)(u32)v, d, 64);
    u32 hi = (u32)__shfl_down((int)(u32)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 shfl_t(u32 v, int src) { return (u32)__shfl((int)v, src, 64); }
__device__ __forceinline__ u64 shfl_t(u64 v, int src)
{
    u32 lo = (u32)__shfl((int)(u32)v, src, 64);
    u32 hi = (u32)__shfl((int)(u32)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}

// inclusive scan across the 64 lanes of a wave (commutative op)
template <typename T, typename Op>
__device__ __forceinline__ T wave_scan_inclusive(T v, Op op)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = shfl_up_t(v, d);
        if (lane >= d) v = op(o, v);
    }
    return v;
}

// Workgroup-wide scan of one value per thread.  NWAVES = blockDim.x / 64.
// Returns the exclusive prefix of the calling thread; *total gets the block total.
// smem must hold NWAVES values; contains two barriers.
template <typename T, typename Op, int NWAVES>
__device__ __forceinline__ T block_scan_exclusive(T v, Op op, T identity, T *smem, T *total)
{
    const int lane = lane_id(), w = wave_id();
    T inc = wave_scan_inclusive(v, op);
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    T wave_prefix = identity, tot = identity;
#pragma unroll
    for (int i = 0; i < NWAVES; i++) {
        T s = smem[i];
        if (i < w) wave_prefix = op(wave_prefix, s);
        tot = op(tot, s);
    }
    __syncthreads();
    T exc = shfl_up_t(inc, 1);
    if (lane == 0) exc = identity;
    *total = tot;
    return op(wave_prefix, exc);
}

// lanes of the wave whose 8-bit digit equals the caller's (restricted to `valid` lanes).
// The radix kernels are VALU-bound and most of their VALU work is this function, so it is written for instruction
// count: per digit bit, t = 0 / -1 from a sign-extended one-bit field, one compare for the ballot, and on each
// 32-bit half an XNOR with the ballot (lanes whose bit equals mine) folded in with an AND -- six instructions.
__device__ __forceinline__ u64 match_digit8(u32 digit, bool valid)
{
    const u64 vm = __ballot(valid);
    u32 plo = (u32)vm, phi = (u32)(vm >> 32);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const u32 t = (u32)__builtin_amdgcn_sbfe((int)digit, (unsigned)b, 1u);      // 0 or ~0
        const u64 m = __ballot((int)t < 0);                                              // sign test of t itself: no second extract
        plo &= ~((u32)m ^ t);
        phi &= ~((u32)(m >> 32) ^ t);
    }
    return ((u64)phi << 32) | plo;
}
