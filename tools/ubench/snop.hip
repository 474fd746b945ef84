// what an `s_nop 0` between two dependent packed fp32 instructions costs (LLVM's gfx940 "dst_sel forwarding" hazard check takes every
// VOP3P instruction with the default op_sel_hi for a partial writer and separates it from its consumer): chains of v_pk_mul_f32 ->
// v_pk_fma_f32 with and without the s_nop, and with an independent instruction in between instead; W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(const float *in, float *out, unsigned long long *clk, int iters)
{
    v2 a[8], w = {in[0], in[1]}, c = {in[2], in[3]};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2{in[4 + i] + threadIdx.x, in[12 + i]};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(c));
                if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(c));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %3, %3, %1\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %3, %3, %1, %2" : "+v"(a[i]), "+v"(a[(i + 1) & 7]) : "v"(w), "v"(c), "v"(a[(i + 1) & 7]));
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) atomicMax(clk + blockIdx.x % 1, t1 - t0);
}

int main()
{
    float *in, *out; unsigned long long *clk;
    hipMalloc(&in, 256); hipMemset(in, 0, 256); hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&clk, 8);
    for (int w = 1; w <= 4; w *= 2)
        for (int mode = 0; mode < 2; ++mode) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(clk, 0, 8);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * w), 0, 0, in, out, clk, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * w), 0, 0, in, out, clk, iters);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
                if (rep) printf("waves/SIMD %d  %s: slowest wave %.2f cycles per (mul, fma) pair; kernel %.1f us -> %.2f cycles per pair and SIMD at 2.1 GHz\n", w, mode ? "with s_nop 0" : "no nop      ",
                                double(c) / (iters * 32.0), ms * 1e3, ms * 1e-3 * 2.1e9 / (iters * 32.0 * w));
            }
        }
    return 0;
}
