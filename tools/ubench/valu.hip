// VALU issue rates on gfx950: cycles per instruction per wave for fma / pk_fma / sqrt, with 1, 2, 4 waves per SIMD and
// 16 or 2 independent dependency chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int OP, int CHAINS>
__global__ void k(float *out, long long *clk, float a, float b)
{
    v2 x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = v2{float(threadIdx.x + i), float(i)};
    v2 av{a, a}, bv{b, b};
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = j % CHAINS;
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c].x) : "v"(a), "v"(b));
                else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(av), "v"(bv));
                else if (OP == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[c].x));
                else if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(av));
                else if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c].x) : "v"(a));
            }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    // workgroup time = last wave's end - first wave's start: the oldest wave of a SIMD gets issue priority, so a
    // wave-0-only clock overstates what four co-resident waves achieve together
    __shared__ long long s0[16], s1[16];
    if ((threadIdx.x & 63) == 0) { s0[threadIdx.x >> 6] = t0; s1[threadIdx.x >> 6] = t1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long a = s0[0], b = s1[0];
        for (int w = 1; w < int(blockDim.x) / 64; ++w) { a = s0[w] < a ? s0[w] : a; b = s1[w] > b ? s1[w] : b; }
        clk[blockIdx.x] = b - a;
    }
}
int main()
{
    float *out; long long *clk;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&clk, 8 * 1024);
    std::vector<long long> h(1024);
    auto run = [&](const char *name, auto kern, int threads) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, clk, 1.0001f, 0.5f); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), clk, 8 * 256, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("%-34s waves/SIMD=%d  %.2f cyc/instr/wave  %.2f cyc/instr/SIMD\n", name, threads / 256, avg / 4096, avg / 4096 / (threads / 256));
    };
    for (int threads : {256, 512, 1024}) {
        run("v_fma_f32 16 chains", k<0, 16>, threads);
        run("v_fma_f32 2 chains", k<0, 2>, threads);
        run("v_fma_f32 1 chain", k<0, 1>, threads);
        run("v_pk_fma_f32 16 chains", k<1, 16>, threads);
        run("v_pk_fma_f32 2 chains", k<1, 2>, threads);
        run("v_pk_add_f32 16 chains", k<3, 16>, threads);
        run("v_sqrt_f32 16 chains", k<2, 16>, threads);
        run("v_cndmask_b32 16 chains", k<4, 16>, threads);
    }
    return 0;
}
