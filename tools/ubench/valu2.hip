// per-opcode VALU issue cost on gfx950 (2 waves / SIMD, 16 independent chains): ticks per instruction per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define BODY(ASM) \
    _Pragma("unroll 1") for (int i = 0; i < 16; ++i) { \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) { ASM; } }
template <int OP>
__global__ void k(float *out, long long *clk, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = float(threadIdx.x + i);
    float2 y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = make_float2(float(threadIdx.x + i), float(i));
    const float2 pa = make_float2(a, b), pb = make_float2(b, a);
    long long t0 = __builtin_readcyclecounter();
    if (OP == 0) BODY(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b)))
    if (OP == 1) BODY(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(a)))
    if (OP == 2) BODY(asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x[j]) : "v"(a) : "s20", "s21"))
    if (OP == 3) BODY(asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x[j]), "v"(a) : "vcc"))
    if (OP == 4) BODY(asm volatile("v_cmp_gt_f32 s[20:21], %0, %1" : : "v"(x[j]), "v"(a) : "s20", "s21"))
    if (OP == 5) BODY(asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 6) BODY(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b)))
    if (OP == 7) BODY(asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 8) BODY(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 9) BODY(asm volatile("v_mov_b32 %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 10) BODY(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(a) : "vcc"))
    if (OP == 11) BODY(asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(x[j])))
    if (OP == 12) BODY(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 13) BODY(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "s"(a)))
    if (OP == 14) BODY(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "s"(a), "v"(b)))
    if (OP == 15) BODY(asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 16) BODY(asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x[j])))
    if (OP == 17) BODY(asm volatile("v_mul_f32 %0, 0x3f7b4a23, %0" : "+v"(x[j])))
    if (OP == 18) BODY(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b)))
    if (OP == 19) BODY(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 20) BODY(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_fma_f32 %2, %2, %1, %1" : "+v"(x[j]), "+v"(a), "+v"(x[(j+1)&15]) : : "s20", "s21"))
    if (OP == 21) BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[j & 7]) : "v"(pa), "v"(pb)))
    if (OP == 22) BODY(asm volatile("v_pk_fma_f32 %0, %0, s[20:21], %1" : "+v"(y[j & 7]) : "v"(pb) : "s20", "s21"))
    if (OP == 23) BODY(asm volatile("v_pk_mul_f32 %0, %0, s[20:21]" : "+v"(y[j & 7]) : : "s20", "s21"))
    if (OP == 24) BODY(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[j & 7]) : "v"(pa)))
    if (OP == 25) BODY(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[j & 7]) : "v"(pa)))
    if (OP == 26) BODY(asm volatile("v_pk_fma_f32 %0, %0, s[20:21], %1 op_sel_hi:[1,0,1]" : "+v"(y[j & 7]) : "v"(pb) : "s20", "s21"))
    if (OP == 27) BODY(asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(x[j]) : "v"(b) : "s20"))
    if (OP == 28) BODY(asm volatile("v_fmac_f32 %0, s20, %1" : "+v"(x[j]) : "v"(b) : "s20"))
    if (OP == 29) BODY(asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[j])))
    if (OP == 30) BODY(asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(x[j])))
    if (OP == 31) BODY(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b)))
    long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += y[i].x + y[i].y;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    // workgroup time = last wave's end - first wave's start (the oldest wave of a SIMD gets issue priority)
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
    int threads = 512;
    auto run = [&](const char *name, auto kern, int n) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, clk, 1.0001f, 0.5f); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), clk, 8 * 256, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("%-44s %.2f ticks/instr/wave   %.2f ticks/instr/SIMD (%d waves/SIMD)\n", name, avg / 4096 / n, avg / 4096 / n / (threads / 256), threads / 256);
    };
    for (threads = 512; threads <= 1024; threads += 512) {
    run("v_fma_f32", k<0>, 1); run("v_cndmask vcc", k<1>, 1); run("v_cndmask s[20:21]", k<2>, 1);
    run("v_cmp_gt_f32 vcc", k<3>, 1); run("v_cmp_gt_f32 s[20:21]", k<4>, 1); run("v_max_f32", k<5>, 1);
    run("v_max3_f32", k<6>, 1); run("v_and_b32", k<7>, 1); run("v_mul_f32", k<8>, 1); run("v_mov_b32", k<9>, 1);
    run("v_cmp vcc + v_cndmask vcc (per instr)", k<10>, 2); run("v_bfe_i32", k<11>, 1); run("v_add_f32 v,v", k<12>, 1);
    run("v_add_f32 v,s", k<13>, 1); run("v_fma_f32 v,s,v", k<14>, 1); run("v_min_u32", k<15>, 1);
    run("v_mul_f32 inline const", k<16>, 1); run("v_mul_f32 literal", k<17>, 1); run("v_fmac_f32", k<18>, 1); run("v_add_u32", k<19>, 1);
    run("cndmask s + fma interleaved (per instr)", k<20>, 2);
    run("v_pk_fma_f32 v,v,v", k<21>, 1); run("v_pk_fma_f32 v,s[2],v", k<22>, 1); run("v_pk_mul_f32 v,s[2]", k<23>, 1);
    run("v_pk_mul_f32 v,v", k<24>, 1); run("v_pk_add_f32 v,v", k<25>, 1); run("v_pk_fma_f32 v,s[2],v op_sel_hi", k<26>, 1);
    run("v_fma_f32 v,s20,v", k<27>, 1); run("v_fmac_f32 s20,v", k<28>, 1); run("v_sqrt_f32", k<29>, 1); run("v_lshlrev_b32", k<30>, 1);
    run("v_add3_u32", k<31>, 1);
    }
    return 0;
}
