// random_read.hip -- how fast can gfx950 serve random 4-byte reads from a 4 GiB table, and does the fetch
// granularity change with the load flavour / allocation kind?  (Experiment behind the inverse walk's design.)
//   hipcc -O3 --offload-arch=gfx950 random_read.hip -o random_read && ./random_read
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u32 mix(u64 z) { z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull; z ^= z >> 33; return (u32)z; }
__global__ void fill(u32 *t, u64 n, u32 mask) { for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) t[i] = mix(i) & mask; }

template <int MODE> __device__ __forceinline__ u32 ld(const u32 *p)
{
    if (MODE == 1) return __builtin_nontemporal_load(p);
    if (MODE == 2) { u32 v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
    if (MODE == 3) { u32 v; asm volatile("global_load_dword %0, %1, off nt sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
    if (MODE == 4) { u32 v; asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
    return *p;
}
// every lane chases `steps` dependent reads
template <int MODE> __global__ __launch_bounds__(256) void chase(const u32 *__restrict__ t, u32 mask, int steps, u32 *out)
{
    u32 x = mix(blockIdx.x * 256ull + threadIdx.x + 12345) & mask;
    for (int s = 0; s < steps; s++) x = ld<MODE>(t + x);
    if (x == 0xffffffffu) out[0] = x;
}
template <int MODE> float run(const u32 *t, u32 mask, u32 *out, int blocks, int steps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    chase<MODE><<<blocks, 256>>>(t, mask, 8, out);
    CK(hipEventRecord(a));
    chase<MODE><<<blocks, 256>>>(t, mask, steps, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30;
    const u64 n = 1ull << log2n; const u32 mask = (u32)(n - 1);
    u32 *out; CK(hipMalloc(&out, 4096));
    for (int kind = 0; kind < 3; kind++) {
        u32 *t = nullptr;
        if (kind == 0) CK(hipMalloc(&t, n * 4));
        else if (kind == 1) { if (hipExtMallocWithFlags((void **)&t, n * 4, hipDeviceMallocUncached) != hipSuccess) { printf("uncached alloc unsupported\n"); continue; } }
        else { if (hipExtMallocWithFlags((void **)&t, n * 4, hipDeviceMallocFinegrained) != hipSuccess) { printf("finegrained alloc unsupported\n"); continue; } }
        fill<<<8192, 256>>>(t, n, mask); CK(hipDeviceSynchronize());
        const int blocks = 2048, steps = 256;
        const double reads = (double)blocks * 256 * steps;
        float ms;
        ms = run<0>(t, mask, out, blocks, steps); printf("alloc %d mode plain      : %7.2f ms  %6.1f G reads/s\n", kind, ms, reads / ms * 1e-6);
        ms = run<1>(t, mask, out, blocks, steps); printf("alloc %d mode nontemporal: %7.2f ms  %6.1f G reads/s\n", kind, ms, reads / ms * 1e-6);
        ms = run<2>(t, mask, out, blocks, steps); printf("alloc %d mode sc0 sc1    : %7.2f ms  %6.1f G reads/s\n", kind, ms, reads / ms * 1e-6);
        ms = run<3>(t, mask, out, blocks, steps); printf("alloc %d mode nt sc1     : %7.2f ms  %6.1f G reads/s\n", kind, ms, reads / ms * 1e-6);
        ms = run<4>(t, mask, out, blocks, steps); printf("alloc %d mode sc0        : %7.2f ms  %6.1f G reads/s\n", kind, ms, reads / ms * 1e-6);
        CK(hipFree(t));
    }
    return 0;
}
