// VALU issue cost with REALISTIC register traffic on gfx950: every instruction reads distinct VGPRs and writes another one
// (valu2.hip / valu3.hip measure d = op(d, a) with one shared source, which hides the operand-fetch limits).  Compiler-generated
// straight-line code over four 16-register arrays; 512-thread workgroups, 2 or 4 waves per SIMD, whole-launch time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
#define ROUND(EXPR_X, EXPR_Y, EXPR_Z, EXPR_W) \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) x[j] = EXPR_X; \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) y[j] = EXPR_Y; \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) z[j] = EXPR_Z; \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) w[j] = EXPR_W;
template <int OP>
__global__ void __launch_bounds__(512, 4) k(float *out, const float *in, int iters)
{
    float x[16], y[16], z[16], w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = in[threadIdx.x + i]; y[i] = in[threadIdx.x + 16 + i]; z[i] = in[threadIdx.x + 32 + i]; w[i] = in[threadIdx.x + 48 + i]; }
    constexpr float K = 0.98078528f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { ROUND(__builtin_fmaf(y[j], z[j], w[j]), __builtin_fmaf(z[j], w[j], x[j]), __builtin_fmaf(w[j], x[j], y[j]), __builtin_fmaf(x[j], y[j], z[j])) }
        if (OP == 1) { ROUND(y[j] + z[j], z[j] + w[j], w[j] + x[j], x[j] + y[j]) }
        if (OP == 2) { ROUND(__builtin_fmaf(y[j], K, z[j]), __builtin_fmaf(z[j], K, w[j]), __builtin_fmaf(w[j], K, x[j]), __builtin_fmaf(x[j], K, y[j])) }
        if (OP == 3) { ROUND(__builtin_fmaf(y[j], K, x[j]), __builtin_fmaf(z[j], K, y[j]), __builtin_fmaf(w[j], K, z[j]), __builtin_fmaf(x[j], K, w[j])) }
        if (OP == 4) { ROUND(__builtin_fmaf(2.f, y[j], -z[j]), __builtin_fmaf(2.f, z[j], -w[j]), __builtin_fmaf(2.f, w[j], -x[j]), __builtin_fmaf(2.f, x[j], -y[j])) }
        if (OP == 5) { ROUND(y[j] * K, z[j] * K, w[j] * K, x[j] * K) }
        if (OP == 6) { ROUND(__builtin_fmaxf(y[j], z[j]), __builtin_fmaxf(z[j], w[j]), __builtin_fmaxf(w[j], x[j]), __builtin_fmaxf(x[j], y[j])) }
        if (OP == 7) {
            v2 *X = reinterpret_cast<v2 *>(x), *Y = reinterpret_cast<v2 *>(y), *Z = reinterpret_cast<v2 *>(z), *W = reinterpret_cast<v2 *>(w);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) X[j] = Y[j] + Z[j];
#pragma unroll
                for (int j = 0; j < 8; ++j) Y[j] = Z[j] + W[j];
#pragma unroll
                for (int j = 0; j < 8; ++j) Z[j] = W[j] + X[j];
#pragma unroll
                for (int j = 0; j < 8; ++j) W[j] = X[j] + Y[j];
            }
        }
        if (OP == 8) { ROUND(__builtin_fmaf(y[j], y[j], w[j]), __builtin_fmaf(z[j], z[j], x[j]), __builtin_fmaf(w[j], w[j], y[j]), __builtin_fmaf(x[j], x[j], z[j])) }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(x[i]), "+v"(y[i]), "+v"(z[i]), "+v"(w[i]));
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += x[i] + y[i] + z[i] + w[i];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int OP>
static void run(const char *name, float *out, const float *in, int waves4)
{
    const int iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves4 / 2), dim3(512), 0, 0, out, in, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double n = double(iters) * 64.0;                       // instructions per wave
    printf("%-36s %d waves/SIMD: %6.2f SIMD clocks per wave-instruction (64 per iteration)\n", name, waves4, best * 1e-3 * 2.4e9 / (n * waves4));
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 1024 * 512 * 4);
    (void)hipMalloc(&in, 8192 * 4);
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = 0.001f * float(i % 97) - 0.04f;
    (void)hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int w4 = 2; w4 <= 4; w4 += 2) {
        run<0>("v_fma_f32 d, a, b, c", out, in, w4);
        run<1>("v_add_f32 d, a, b", out, in, w4);
        run<2>("fma(a, K, b) (fmamk)", out, in, w4);
        run<3>("fma(a, K, d) (fmac K)", out, in, w4);
        run<4>("fma(2, a, -b)", out, in, w4);
        run<5>("v_mul_f32 d, K, a", out, in, w4);
        run<6>("v_max_f32 d, a, b", out, in, w4);
        run<7>("v_pk_add_f32 (32 per iteration x2)", out, in, w4);
        run<8>("fma(a, a, b)", out, in, w4);
    }
    return 0;
}
