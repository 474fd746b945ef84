// radix-32 butterflies, per-wave issue cost: packed DIF (fft_common.hpp difPacked) against the scalar-FMA DIT form (fft_scalar.hpp),
// with and without the inter-pass twiddles.  512-thread workgroups, two per CU, time of the whole launch / iterations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../signalizer_amd/csrc/fft_common.hpp"
#include "../../signalizer_amd/csrc/fft_scalar.hpp"
using namespace sgz;

template <int MODE>
__global__ void __launch_bounds__(512, 4) k(float *out, const float *in, int iters)
{
    constexpr int R = 32;
    __shared__ float2 tab[32 * 32];
    for (int i = threadIdx.x; i < 1024; i += 512) tab[i] = make_float2(in[2048 + 2 * i], in[2049 + 2 * i]);
    __syncthreads();
    v2 c[R];
#pragma unroll
    for (int i = 0; i < R; ++i) c[i] = v2{in[threadIdx.x + i], in[threadIdx.x + i + 64]};
    TwFactors<5> tw;
    if (MODE == 2) tw.load(reinterpret_cast<const float2 *>(in + 1024), threadIdx.x & 31, 32);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) difPacked<R, R, 0>(c);
        if (MODE == 1) ditScalar<5, 0>(c);
        if (MODE == 4) ditPacked<5, 0>(c);
        if (MODE == 2) { difPacked<R, R, 0>(c); tw.apply(c); }
        if (MODE == 3) {
            ditScalar<5, 0>(c);
#pragma unroll
            for (int q = 1; q < R; ++q) {
                const float2 w = tab[q * 32 + (threadIdx.x & 31)];
                c[brev(q, 5)] = cmulScalar(c[brev(q, 5)], w);
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) asm volatile("" : "+v"(c[i]));
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) acc += c[i].x + c[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char *name, float *out, const float *in)
{
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(512), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 2 workgroups x 8 waves = 4 waves per SIMD; clocks per wave-iteration of SIMD time = t * f / (iters * 4)
        if (rep == 2) printf("%-28s %8.1f us  -> %7.0f SIMD clocks per wave per iteration (2.4 GHz)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / (iters * 4.0));
    }
}

int main()
{
    float *out, *in;
    hipMalloc(&out, 512 * 512 * 4);
    hipMalloc(&in, 8192 * 4);
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = 0.001f * float(i % 97) - 0.04f;
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    run<0>("difPacked<32>", out, in);
    run<1>("ditScalar<5>", out, in);
    run<2>("difPacked + TwFactors", out, in);
    run<3>("ditScalar + 31 cmul (LDS)", out, in);
    run<4>("ditPacked<5>", out, in);
    // numerics: the three forms on the same input
    {
        std::vector<float> a(512 * 512), b(512 * 512), c2(512 * 512);
        hipLaunchKernelGGL(k<0>, dim3(512), dim3(512), 0, 0, out, in, 1); hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k<1>, dim3(512), dim3(512), 0, 0, out, in, 1); hipMemcpy(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k<4>, dim3(512), dim3(512), 0, 0, out, in, 1); hipMemcpy(c2.data(), out, c2.size() * 4, hipMemcpyDeviceToHost);
        double d1 = 0, d2 = 0, mx = 0;
        for (size_t i = 0; i < a.size(); ++i) { d1 = fmax(d1, fabs(a[i] - b[i])); d2 = fmax(d2, fabs(a[i] - c2[i])); mx = fmax(mx, fabs(a[i])); }
        printf("checksum differences vs difPacked: scalar DIT %.3g, packed DIT %.3g (max |sum| %.3g)\n", d1, d2, mx);
    }
    return 0;
}
