// v_mfma_f32_32x32x16_bf16 issue cadence per SIMD: one dependent chain (every product accumulates into the previous one's result), two
// and four alternating chains, and each with four independent v_fmac between the products; W waves per SIMD.  Cycles per MFMA from
// s_memtime of wave 0 over a long loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int FILL>
__global__ void k(const uint4 *in, float *out, unsigned long long *clk, int iters)
{
    bf16x8 a = __builtin_bit_cast(bf16x8, in[threadIdx.x & 63]), b = __builtin_bit_cast(bf16x8, in[64 + (threadIdx.x & 63)]);
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x16{0};
    float f[4] = {1.f, 2.f, 3.f, 4.f};
    const float g = out[0];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CHAINS], 0, 0, 0);
            asm volatile("" : "+v"(acc[m % CHAINS]));
            if (FILL) {
#pragma unroll
                for (int q = 0; q < FILL; ++q) f[q & 3] = __builtin_fmaf(f[q & 3], g, f[(q + 1) & 3]);
                asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = f[0] + f[1] + f[2] + f[3];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int CHAINS, int FILL>
void run(const uint4 *in, float *out, unsigned long long *clk, int wavesPerSimd)
{
    const int iters = 2000;
    // one workgroup of 4 * W waves per CU, 256 workgroups
    hipLaunchKernelGGL((k<CHAINS, FILL>), dim3(256), dim3(256 * wavesPerSimd), 0, 0, in, out, clk, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<CHAINS, FILL>), dim3(256), dim3(256 * wavesPerSimd), 0, 0, in, out, clk, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("chains %d fill %d waves/SIMD %d: %.1f cycles per MFMA per wave, %.1f per SIMD\n", CHAINS, FILL, wavesPerSimd, double(c) / (iters * 24.0), double(c) / (iters * 24.0) / wavesPerSimd);
}

int main()
{
    uint4 *in; float *out; unsigned long long *clk;
    hipMalloc(&in, 128 * 16); hipMemset(in, 0, 128 * 16);
    hipMalloc(&out, 4 * (1 + 256 * 1024)); hipMemset(out, 0, 4 * (1 + 256 * 1024));
    hipMalloc(&clk, 8);
    for (int w = 1; w <= 3; ++w) {
        run<1, 0>(in, out, clk, w); run<2, 0>(in, out, clk, w); run<4, 0>(in, out, clk, w);
        run<1, 4>(in, out, clk, w); run<2, 4>(in, out, clk, w); run<1, 6>(in, out, clk, w); run<2, 6>(in, out, clk, w); run<1, 8>(in, out, clk, w);
    }
    return 0;
}
