// radix-32 DIF butterflies: scalar (two sets one after the other) vs packed (v2: both sets per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../signalizer_amd/csrc/fft_common.hpp"
using namespace sgz;
// complex-interleaved packed DIF: c[i] = (re, im); twiddle W = (c, -s): d * W = d * (c, c) + (d.y, -d.x) * (s, s)
template <int R, int LEN, int BASE>
__device__ __forceinline__ void difc(v2 (&c)[R])
{
    constexpr int H = LEN / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int a = BASE + i, b = BASE + i + H;
        const v2 x = c[a], y = c[b];
        c[a] = x + y;
        const v2 d = x - y;
        const int j = i * (32 / LEN);
        if (j == 0) c[b] = d;
        else if (j == 8) c[b] = v2{d.y, -d.x};
        else {
            const float cs = cos32(j), sn = sin32(j);
            c[b] = d * v2{cs, cs} + v2{d.y, -d.x} * v2{sn, sn};
        }
    }
    if constexpr (LEN > 2) {
        difc<R, H, BASE>(c);
        difc<R, H, BASE + H>(c);
    }
}

template <int MODE>
__global__ void __launch_bounds__(MODE >= 2 ? 1024 : 512) k(float *out, long long *clk, const float *in)
{
    constexpr int R = 32;
    float acc = 0.f;
    long long t0, t1;
    if (MODE == 0) {
        float reA[R], imA[R], reB[R], imB[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { reA[i] = in[threadIdx.x + i]; imA[i] = in[threadIdx.x + i + 64]; reB[i] = in[threadIdx.x + i + 128]; imB[i] = in[threadIdx.x + i + 192]; }
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            dif<float, R, R, 0>(reA, imA);
            dif<float, R, R, 0>(reB, imB);
#pragma unroll
            for (int i = 0; i < R; ++i) asm volatile("" : "+v"(reA[i]), "+v"(imA[i]), "+v"(reB[i]), "+v"(imB[i]));
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < R; ++i) acc += reA[i] + imA[i] + reB[i] + imB[i];
    } else if (MODE == 2) {          // one set per thread (V = 1): launch with 1024 threads = 4 waves / SIMD
        float reA[R], imA[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { reA[i] = in[threadIdx.x + i]; imA[i] = in[threadIdx.x + i + 64]; }
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            dif<float, R, R, 0>(reA, imA);
#pragma unroll
            for (int i = 0; i < R; ++i) asm volatile("" : "+v"(reA[i]), "+v"(imA[i]));
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < R; ++i) acc += reA[i] + imA[i];
    } else if (MODE == 3) {          // V = 1, complex-interleaved packed math, 1024 threads
        v2 c[R];
#pragma unroll
        for (int i = 0; i < R; ++i) c[i] = v2{in[threadIdx.x + i], in[threadIdx.x + i + 64]};
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            difc<R, R, 0>(c);
#pragma unroll
            for (int i = 0; i < R; ++i) asm volatile("" : "+v"(c[i]));
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < R; ++i) acc += c[i].x + c[i].y;
    } else if (MODE == 4) {          // V = 1, packed complex with explicit op_sel (fft_common.hpp difPacked), 1024 threads
        v2 c[R];
#pragma unroll
        for (int i = 0; i < R; ++i) c[i] = v2{in[threadIdx.x + i], in[threadIdx.x + i + 64]};
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            difPacked<R, R, 0>(c);
#pragma unroll
            for (int i = 0; i < R; ++i) asm volatile("" : "+v"(c[i]));
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < R; ++i) acc += c[i].x + c[i].y;
    } else if (MODE == 5) {          // correctness: scalar vs packed on the same input, max |diff| -> out
        float re[R], im[R];
        v2 c[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { re[i] = in[threadIdx.x + i]; im[i] = in[threadIdx.x + i + 64]; c[i] = v2{re[i], im[i]}; }
        t0 = t1 = 0;
        dif<float, R, R, 0>(re, im);
        difPacked<R, R, 0>(c);
        TwFactors<5> tw;
        tw.load(reinterpret_cast<const float2 *>(in + 1024), threadIdx.x & 31, 32);
        tw.apply(re, im);
        tw.apply(c);
#pragma unroll
        for (int i = 0; i < R; ++i) acc = fmaxf(acc, fmaxf(fabsf(re[i] - c[i].x), fabsf(im[i] - c[i].y)));
    } else {
        v2 re[R], im[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { re[i] = v2{in[threadIdx.x + i], in[threadIdx.x + i + 128]}; im[i] = v2{in[threadIdx.x + i + 64], in[threadIdx.x + i + 192]}; }
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            dif<v2, R, R, 0>(re, im);
#pragma unroll
            for (int i = 0; i < R; ++i) asm volatile("" : "+v"(re[i]), "+v"(im[i]));
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < R; ++i) acc += re[i].x + im[i].x + re[i].y + im[i].y;
    }
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
    float *out, *in; long long *clk;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&in, 4 * 4096); hipMalloc(&clk, 8 * 1024);
    hipMemset(in, 0, 4 * 4096);
    std::vector<long long> h(1024);
    auto run = [&](const char *name, auto kern, int threads = 512) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, clk, in); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), clk, 8 * 256, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("%-34s %.0f ticks per (2 x 32-point DIF)\n", name, avg / 16);
    };
    run("scalar A then B", k<0>);
    run("packed v2", k<1>);
    run("scalar V=1, 1024 threads", k<2>, 1024);
    run("packed complex V=1, 1024 threads", k<3>, 1024);
    run("packed complex, explicit op_sel, 1024 thr", k<4>, 1024);
    {
        std::vector<float> hin(4096);
        for (int i = 0; i < 4096; ++i) hin[i] = float((i * 2654435761u) % 2001) / 1000.f - 1.f;
        hipMemcpy(in, hin.data(), 4 * 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k<5>, dim3(1), dim3(1024), 0, 0, out, clk, in); hipDeviceSynchronize();
        std::vector<float> ho(1024); hipMemcpy(ho.data(), out, 4 * 1024, hipMemcpyDeviceToHost);
        float m = 0; for (float v : ho) m = v > m ? v : m;
        printf("scalar vs packed DIF + twiddles: max |diff| = %g\n", m);
    }
    return 0;
}
