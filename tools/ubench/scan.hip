// The arg-max piece scan of K_A's mapping (16 csf values from LDS per piece, first-strictly-greater winner) in isolation:
// ticks for 4 pieces per thread with 1, 2, 4 waves per SIMD, for different formulations of the per-element update.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int VARIANT>
__global__ void k(const uint32_t *items, uint2 *out, long long *clk, int reps)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 35000; i += blockDim.x) lds[i] = float((i * 2654435761u) % 1000) * 1e-3f;
    __syncthreads();
    uint32_t accK = 0, accB = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; ++rep) {
        uint32_t iw[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) iw[b] = items[(rep * 4 + b) * blockDim.x + tid];
        float mv[4][16];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k0 = int(iw[b] & 0xFFFFu) << 4;
            const float *src = lds + (k0 + (k0 >> 5));        // the kernel's bank-padded layout
#pragma unroll
            for (int j = 0; j < 16; ++j) mv[b][j] = src[j];
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k0 = int(iw[b] & 0xFFFFu) << 4;
            const int lo = int((iw[b] >> 16) & 15u), hi = int((iw[b] >> 20) & 15u);
            const uint32_t mask = (0xFFFFu >> (15 - hi)) & (0xFFFFu << lo);
            const uint32_t r = (iw[b] >> 24) & 1u;
            uint32_t best = r, bestK = 0xFFFFFFFFu;
            if (VARIANT == 0) {                  // as in the kernel: cmp + max + cndmask
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float sq = mv[b][j] * mv[b][j];
                    const uint32_t keep = uint32_t(__builtin_amdgcn_sbfe(int(mask), j, 1));
                    const uint32_t sqm = __float_as_uint(sq) & keep;
                    const bool take = sqm + r > best;
                    best = sqm > best ? sqm : best;
                    bestK = take ? uint32_t(k0 + j) : bestK;
                }
            } else if (VARIANT == 1) {           // two-step: max tree first, then first / last index equal to the max
                uint32_t sqm[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float sq = mv[b][j] * mv[b][j];
                    const uint32_t keep = uint32_t(__builtin_amdgcn_sbfe(int(mask), j, 1));
                    sqm[j] = __float_as_uint(sq) & keep;
                }
                uint32_t m = r;
#pragma unroll
                for (int j = 0; j < 16; ++j) m = sqm[j] > m ? sqm[j] : m;
                // left: first j with sqm == m ; right: last j with sqm == m  (m > 0 guaranteed iff any valid value > 0 / >= r)
                uint32_t idx = 0xFFFFFFFFu;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int j = 15 - jj;       // descending, so the lowest j wins for the left side
                    idx = (sqm[j] == m) ? uint32_t(j) : idx;
                }
                uint32_t idxR = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < 16; ++j) idxR = (sqm[j] == m) ? uint32_t(j) : idxR;
                const uint32_t w = r ? idxR : idx;
                best = m;
                bestK = (m > r - (r & 1u) * 0u && m != r) ? uint32_t(k0) + w : 0xFFFFFFFFu;
            } else {                             // packed key: (bits & ~15) | (15 - j) -- not exact, speed reference only
                uint32_t m = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float sq = mv[b][j] * mv[b][j];
                    const uint32_t keep = uint32_t(__builtin_amdgcn_sbfe(int(mask), j, 1));
                    const uint32_t key = ((__float_as_uint(sq) & keep) & ~15u) | uint32_t(15 - j);
                    m = key > m ? key : m;
                }
                best = m & ~15u;
                bestK = uint32_t(k0) + 15u - (m & 15u);
            }
            accK += bestK; accB ^= best;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = make_uint2(accK, accB);
    __shared__ long long s0[16], s1[16];
    if ((tid & 63) == 0) { s0[tid >> 6] = t0; s1[tid >> 6] = t1; }
    __syncthreads();
    if (tid == 0) {
        long long a = s0[0], b = s1[0];
        for (int w = 1; w < int(blockDim.x) / 64; ++w) { a = s0[w] < a ? s0[w] : a; b = s1[w] > b ? s1[w] : b; }
        clk[blockIdx.x] = b - a;
    }
}
int main()
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 35000 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 35000 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 35000 * 4);
    const int reps = 8;
    std::vector<uint32_t> h(reps * 4 * 1024);
    for (size_t i = 0; i < h.size(); ++i) { uint32_t w = uint32_t(i % 2048),      // consecutive windows for consecutive lanes, as the plan lists them
             lo = (i * 31u) % 8u, hi = 8u + (i * 17u) % 8u; h[i] = w | (lo << 16) | (hi << 20) | ((i & 1u) << 24); }
    uint32_t *items; uint2 *out; long long *clk;
    hipMalloc(&items, h.size() * 4); hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&clk, 8 * 256);
    hipMemcpy(items, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<long long> c(256);
    auto run = [&](const char *name, auto kern, int threads) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 35000 * 4, 0, items, out, clk, reps); hipDeviceSynchronize(); }
        hipMemcpy(c.data(), clk, 8 * 256, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
        printf("%-40s %d waves/SIMD: %7.0f ticks per batch of 4 pieces per thread (whole workgroup)\n", name, threads / 256, avg / reps);
    };
    for (int threads : {256, 512, 1024}) {
        run("cmp + max + cndmask (kernel)", k<0>, threads);
        run("max tree, then index of the max", k<1>, threads);
        run("packed key (inexact, reference)", k<2>, threads);
    }
    return 0;
}
