// Hardware probe (MI355X, gfx950): does a 64-bit VALU shift read its shift amount correctly when the amount sits in the
// wave's LAST allocated VGPR (v15 of a 16-register wave) while other waves share the SIMD?   See tools/check_shift64.py.
// build: hipcc --offload-arch=gfx950 -O3 -o shift64_probe tools/probes/shift64_probe.hip        run: ./shift64_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef uint64_t u64;
typedef uint32_t u32;

// REG = 15: the amount is in the last register of the allocation (the wave has 16 VGPRs);  REG = 14: it is not;
// REG = 15 and v16 clobbered: 17 VGPRs, the allocation goes on behind the amount
#define PROBE(NAME, REG, ...)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(u64 x, int single_lane, unsigned long long *bad, u64 *first)          \
    {                                                                                                                  \
        if (single_lane && (threadIdx.x & 63)) return;                                                                 \
        const u32 gid = blockIdx.x * 256 + threadIdx.x;                                                                \
        u32 wrong = 0;                                                                                                 \
        u64 got_bad = 0; u32 amt_bad = 0;                                                                              \
        _Pragma("unroll 1") for (u32 it = 0; it < 64; it++) {                                                          \
            const u32 a = (gid * 7 + it * 13) & 31;                                                                    \
            u64 r;                                                                                                     \
            asm volatile("v_mov_b32 v" #REG ", %1\n\ts_nop 4\n\tv_lshlrev_b64 %0, v" #REG ", %2\n\ts_nop 1"           \
                         : "=v"(r) : "v"(a), "v"(x) : "v" #REG __VA_ARGS__);                                          \
            if (r != (x << a)) { if (!wrong) { got_bad = r; amt_bad = a; } wrong++; }                                  \
        }                                                                                                              \
        if (wrong) {                                                                                                   \
            if (atomicAdd(bad, (unsigned long long)wrong) == 0) { first[0] = got_bad; first[1] = amt_bad; first[2] = gid; } \
        }                                                                                                              \
    }
PROBE(probe_last, 15)
PROBE(probe_inner, 14)
PROBE(probe_padded, 15, , "v16")      // the workaround used in the library: the amount in v15, one more register allocated behind it

static void run(const char *name, void (*k)(u64, int, unsigned long long *, u64 *), int single, unsigned blocks)
{
    unsigned long long *d_bad; u64 *d_first;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 24);
    hipMemset(d_bad, 0, 8); hipMemset(d_first, 0, 24);
    const u64 x = 0xC78F1E3ull;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, x, single, d_bad, d_first);
    hipDeviceSynchronize();
    unsigned long long bad = 0; u64 first[3] = {0, 0, 0};
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(first, d_first, 24, hipMemcpyDeviceToHost);
    const unsigned long long total = (unsigned long long)blocks * (single ? 4 : 256) * 64;
    printf("%-12s %-12s blocks %6u: %llu wrong of %llu shifts", name, single ? "lane 0 only" : "all lanes", blocks, bad, total);
    if (bad) printf("   first: thread %llu amount %llu expected %016llx got %016llx", (unsigned long long)first[2], (unsigned long long)first[1],
                    (unsigned long long)(x << first[1]), (unsigned long long)first[0]);
    printf("\n");
    hipFree(d_bad); hipFree(d_first);
}

int main()
{
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void *)probe_last);  printf("probe_last:  %d VGPRs\n", fa.numRegs);
    hipFuncGetAttributes(&fa, (const void *)probe_inner); printf("probe_inner: %d VGPRs\n", fa.numRegs);
    hipFuncGetAttributes(&fa, (const void *)probe_padded); printf("probe_padded: %d VGPRs\n", fa.numRegs);
    for (int single = 0; single < 2; single++)
        for (unsigned blocks : {64u, 256u, 4096u, 65536u}) {
            run("amount v15", probe_last, single, blocks);
            run("amount v14", probe_inner, single, blocks);
            run("v15 + v16", probe_padded, single, blocks);
        }
    return 0;
}
