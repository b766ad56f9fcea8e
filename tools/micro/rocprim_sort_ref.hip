// rocprim_sort_ref.hip -- measuring stick only (never linked into the engine): how long does the vendor library's radix
// sort take for the engine's round-0 shape, (u64 key, u32 value) pairs, low `bits` key bits, n = 2^log2n ?
//   hipcc -O3 --offload-arch=gfx950 rocprim_sort_ref.hip -o rocprim_sort_ref && ./rocprim_sort_ref 30 40
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u64 mix(u64 z) { z ^= z >> 33; z *= 0xff51afd7ed558ccdull; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ull; z ^= z >> 33; return z; }
__global__ void fill(u64 *k, u32 *v, u64 n, u64 mask) { for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) { k[i] = mix(i) & mask; v[i] = (u32)i; } }
int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30, bits = argc > 2 ? atoi(argv[2]) : 40;
    const u64 n = 1ull << log2n;
    u64 *k0, *k1; u32 *v0, *v1;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    size_t tb = 0; void *tmp = nullptr;
    CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, k1, v0, v1, (int)n, 0, bits));
    CK(hipMalloc(&tmp, tb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        fill<<<8192, 256>>>(k0, v0, n, bits >= 64 ? ~0ull : (1ull << bits) - 1);
        CK(hipEventRecord(a));
        CK(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, k1, v0, v1, (int)n, 0, bits));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("rocprim SortPairs u64/u32 n=2^%d bits=%d temp=%.1f MB : %.2f ms\n", log2n, bits, tb / 1e6, ms);
    }
    return 0;
}
