// gen.hip -- harness utilities: synthetic inputs of SURVEY.md 8(d) generated in device memory
// (splitmix64 is seekable, so every thread computes its own outputs), and a device memcmp.
#include "internal.h"
#include "device_utils.h"

__device__ __forceinline__ u64 splitmix_at(u64 seed, u64 k)
{
    u64 z = seed + k * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__constant__ u8 c_zipf_alpha[96] = {
    ' ', 'e', 't', 'a', 'o', 'i', 'n', 's', 'h', 'r', 'd', 'l', 'c', 'u', 'm', 'w', 'f', 'g', 'y', 'p', 'b', 'v', 'k', 'j', 'x', 'q', 'z',
    '\n', '<', '>', '/', '=', '"', '[', ']', '|', '\'', '.', ',', ':', ';', '-', '_', '(', ')', '&', '#',
    '0', '1', '2', '3', '4', '5', '6', '7', '8', '9',
    'E', 'T', 'A', 'O', 'I', 'N', 'S', 'H', 'R', 'D', 'L', 'C', 'U', 'M', 'W', 'F', 'G', 'Y', 'P', 'B', 'V', 'K', 'J', 'X', 'Q', 'Z',
    '!', '$', '%', '*', '+', '?', '@', '\\', '^', '`', '{', '}', '~'};

// kind 3 ("text"): the zipf stream with back-references; a position is redirected through the largest copy window that
// holds it until it lies in none.  Same integer definition as the CPU generator the tests compare against.
#define TEXT_LEVEL_SALT 0xD1B54A32D192ED03ull
__device__ __forceinline__ u64 text_resolve(u64 seed, u64 p)
{
    int l = 16;
    while (l >= 4) {
        const u64 w = p >> l;
        if (w) {
            const u64 h = splitmix_at(seed + (u64)l * TEXT_LEVEL_SALT, w);
            const u64 mask = l < 8 ? 7u : l < 12 ? 15u : 31u;
            if ((h & mask) == 0) {
                const u64 span = ((w - 1) << l) + 1;
                p = (h >> 8) % span + (p & ((1ull << l) - 1ull));
                l = 16;
                continue;
            }
        }
        l--;
    }
    return p;
}

// each thread produces 8 consecutive bytes
__global__ __launch_bounds__(256) void generate_kernel(int kind, u64 seed, u64 n, u8 *__restrict__ out)
{
    __shared__ u64 cum[96];
    __shared__ u8 alpha[96];
    if (threadIdx.x < 96) alpha[threadIdx.x] = c_zipf_alpha[threadIdx.x];
    if (threadIdx.x == 0) {
        u64 W = 0;
        for (int k = 0; k < 96; k++) { W += (1u << 24) / (u64)(k + 1); cum[k] = W; }
    }
    __syncthreads();
    const u64 W = cum[95];
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q * 8 < n; q += (u64)gridDim.x * 256) {
        const u64 o = q * 8;
        u64 word = 0;
        if (kind == 0) {
            word = splitmix_at(seed, q + 1);
        } else if (kind == 2) {
            const u64 z = splitmix_at(seed, o / 32 + 1) >> (2 * (o % 32));
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const u32 c = (u32)(z >> (2 * j)) & 3u;
                const u8 sym = c == 0 ? 'A' : c == 1 ? 'C' : c == 2 ? 'G' : 'T';
                word |= (u64)sym << (8 * j);
            }
        } else {
            for (int j = 0; j < 8; j++) {
                const u64 z = splitmix_at(seed, (kind == 3 ? text_resolve(seed, o + j) : o + j) + 1);
                const u64 u = ((z >> 32) * W) >> 32;
                int lo = 0, hi = 95;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (u < cum[mid]) hi = mid; else lo = mid + 1; }
                word |= (u64)alpha[lo] << (8 * j);
            }
        }
        if (o + 8 <= n && (((uintptr_t)out) & 7) == 0) {
            *(u64 *)(out + o) = word;
        } else {
            for (int j = 0; j < 8 && o + j < n; j++) out[o + j] = (u8)(word >> (8 * j));
        }
    }
}

int generate_device_impl(bwts_ctx *ctx, int kind, u64 seed, u64 n, u8 *d_out)
{
    if (kind < 0 || kind > 3) return BWTS_E_ARG;
    u64 blocks = (n / 8 + 256) / 256;
    if (blocks > 8192) blocks = 8192;
    generate_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(kind, seed, n, d_out);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(ctx->stream));
    return BWTS_OK;
}

__global__ __launch_bounds__(256) void differ_kernel(const u8 *__restrict__ a, const u8 *__restrict__ b, u64 bytes, unsigned long long *__restrict__ diff)
{
    u32 d = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < bytes; i += (u64)gridDim.x * 256) d |= (u32)(a[i] != b[i]);
    if (__ballot(d) && lane_id() == 0) atomicAdd(diff, 1ull);
}

int device_equal_impl(bwts_ctx *ctx, const u8 *a, const u8 *b, u64 bytes, int *equal)
{
    unsigned long long *diff = (unsigned long long *)(ctx->d_small + 4000);
    HIPC(hipMemsetAsync(diff, 0, sizeof(u64), ctx->stream));
    if (bytes) {
        u64 blocks = (bytes + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        differ_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>(a, b, bytes, diff);
        HIPC(hipGetLastError());
    }
    BWTS_TRY(read_small(ctx, 4000, 1));
    *equal = ctx->h_small[4000] == 0;
    return BWTS_OK;
}
