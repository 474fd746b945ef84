// Does straight-line code that runs ONCE per workgroup (the FFT kernel: ~63 KB of text) pay for instruction fetch?
// Same VALU work as (a) a fully unrolled body of NI fma instructions and (b) a loop over a 256-instruction body.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NI, bool UNROLL>
__global__ void __launch_bounds__(1024) k(float *out, long long *clk, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = float(threadIdx.x + i);
    long long t0 = __builtin_readcyclecounter();
    if (UNROLL) {
#pragma unroll
        for (int i = 0; i < NI / 16; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < NI / 256; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; long long *clk;
    hipMalloc(&out, 4 * 512 * 1024); hipMalloc(&clk, 8 * 1024);
    std::vector<long long> h(1024);
    int threads = 512;
    auto run = [&](const char *name, auto kern, int nwg, int ni) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), 0, 0, out, clk, 1.0001f, 0.5f);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipMemcpy(h.data(), clk, 8 * nwg, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < nwg; ++i) avg += h[i]; avg /= nwg;
        printf("%-28s wgs=%4d threads=%4d wave0 cycles=%8.0f (%.2f cyc/instr/wave, %.2f /SIMD)  kernel %.1f us\n", name, nwg, threads, avg, avg / ni, avg / ni / (threads / 256), best * 1e3);
    };
    for (int th : {512, 1024}) { threads = th;
    for (int nwg : {1, 256}) {
        run("unrolled 8192 fma (64 KB)", k<8192, true>, nwg, 8192);
        run("looped   8192 fma", k<8192, false>, nwg, 8192);
        run("unrolled 4096 fma (32 KB)", k<4096, true>, nwg, 4096);
        run("looped   4096 fma", k<4096, false>, nwg, 4096);
    } }
    return 0;
}
