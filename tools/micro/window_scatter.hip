// window_scatter.hip -- out[idx] = val with idx random inside consecutive windows of W bytes: how much locality does a
// random 4-byte scatter need on this chip before it runs at streaming speed?  (Behind the binned rank-array build.)
//   hipcc -O3 --offload-arch=gfx950 window_scatter.hip -o window_scatter && ./window_scatter
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u32 mix(u64 z) { z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull; z ^= z >> 33; return (u32)z; }
// element i writes to window (i / wlen), at a pseudo-random (bijective within the window: odd multiplier) slot
__global__ __launch_bounds__(256) void scatter(u32 *__restrict__ out, u64 n, int wlog)
{
    const u64 wmask = (1ull << wlog) - 1;
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        const u64 base = i & ~wmask;
        const u64 off = ((i & wmask) * 0x9E3779B1ull + 12345) & wmask;     // permutation of the window
        out[base + off] = (u32)i;
    }
}
__global__ __launch_bounds__(256) void gather(const u32 *__restrict__ in, u32 *__restrict__ out, u64 n, int wlog)
{
    const u64 wmask = (1ull << wlog) - 1;
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        const u64 base = i & ~wmask;
        const u64 off = ((i & wmask) * 0x9E3779B1ull + 12345) & wmask;
        out[i] = in[base + off];
    }
}
int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 28;
    const u64 n = 1ull << log2n;
    u32 *out, *out2; CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&out2, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int wlog = 10; wlog <= log2n; wlog += 2) {
        float ms[2];
        for (int mode = 0; mode < 2; mode++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(a));
                if (mode == 0) scatter<<<8192, 256>>>(out, n, wlog); else gather<<<8192, 256>>>(out, out2, n, wlog);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&ms[mode], a, b));
            }
        }
        printf("n=2^%d window %8.2f MB : scatter %7.2f ms (%5.1f G/s)   gather %7.2f ms (%5.1f G/s)\n", log2n, (double)(4ull << wlog) / 1e6,
               ms[0], n / ms[0] * 1e-6, ms[1], n / ms[1] * 1e-6);
    }
    return 0;
}
