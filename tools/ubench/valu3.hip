// integer / address opcodes on gfx950 (same harness as valu2.hip): ticks per instruction per wave and per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define BODY(ASM) \
    _Pragma("unroll 1") for (int i = 0; i < 16; ++i) { \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) { ASM; } }
template <int OP>
__global__ void k(float *out, long long *clk, unsigned a, unsigned b)
{
    unsigned x[16];
    unsigned long long z[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = threadIdx.x + i;
    long long t0 = __builtin_readcyclecounter();
    if (OP == 0) BODY(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(z[j & 7]) : "v"(a), "v"(b) : "vcc"))
    if (OP == 1) BODY(asm volatile("v_mad_u64_u32 %0, s[20:21], s22, 12, %0" : "+v"(z[j & 7]) : : "s20", "s21", "s22"))
    if (OP == 2) BODY(asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(z[j & 7]) : "v"(z[(j + 1) & 7])))
    if (OP == 3) BODY(asm volatile("v_lshl_add_u64 %0, %0, 2, s[20:21]" : "+v"(z[j & 7]) : : "s20", "s21"))
    if (OP == 4) BODY(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 5) BODY(asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 6) BODY(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[j]) : "v"(a), "v"(b)))
    if (OP == 7) BODY(asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(x[j]), "+v"(a), "+v"(x[(j + 1) & 15]) : : "vcc"))
    if (OP == 8) BODY(asm volatile("v_ashrrev_i32 %0, 5, %0" : "+v"(x[j])))
    if (OP == 9) BODY(asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[j])))
    if (OP == 10) BODY(asm volatile("v_add_u32 %0, 0xfffff7c0, %0" : "+v"(x[j])))
    if (OP == 11) BODY(asm volatile("v_add_u32 %0, s20, %0" : "+v"(x[j]) : : "s20"))
    if (OP == 12) BODY(asm volatile("v_and_b32 %0, 0x7c, %0" : "+v"(x[j])))
    if (OP == 13) BODY(asm volatile("v_mov_b32 %0, 0x210" : "=v"(x[j])))
    if (OP == 14) BODY(asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x[j]) : "s20"))
    if (OP == 15) BODY(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[j])))
    if (OP == 16) BODY(asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7b4a23" : "+v"(x[j]) : "v"(a)))
    if (OP == 17) BODY(asm volatile("v_fmamk_f32 %0, %0, 0x3f7b4a23, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 18) BODY(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[j]) : "v"(a)))
    if (OP == 19) BODY(asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(x[j])))
    long long t1 = __builtin_readcyclecounter();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += unsigned(z[i]) + unsigned(z[i] >> 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = float(acc);
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
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, clk, 3u, 5u); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), clk, 8 * 256, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("%-44s %.2f ticks/instr/wave   %.2f ticks/instr/SIMD (%d waves/SIMD)\n", name, avg / 4096 / n, avg / 4096 / n / (threads / 256), threads / 256);
    };
    for (threads = 512; threads <= 1024; threads += 512) {
        run("v_mad_u64_u32 v,v,v", k<0>, 1); run("v_mad_u64_u32 s,12,v", k<1>, 1); run("v_lshl_add_u64 v,2,v", k<2>, 1);
        run("v_lshl_add_u64 v,2,s[2]", k<3>, 1); run("v_mul_lo_u32", k<4>, 1); run("v_lshl_add_u32", k<5>, 1); run("v_mad_u32_u24", k<6>, 1);
        run("v_add_co + v_addc_co (per instr)", k<7>, 2); run("v_ashrrev_i32", k<8>, 1); run("v_cvt_f32_i32", k<9>, 1);
        run("v_add_u32 literal", k<10>, 1); run("v_add_u32 sgpr", k<11>, 1); run("v_and_b32 inline const", k<12>, 1);
        run("v_mov_b32 literal", k<13>, 1); run("v_readlane_b32", k<14>, 1); run("v_mov_b32_dpp quad_perm", k<15>, 1);
        run("v_fma_f32 v,v,literal", k<16>, 1); run("v_fmamk_f32", k<17>, 1); run("v_sub_f32", k<18>, 1); run("v_xor_b32 literal", k<19>, 1);
    }
    return 0;
}
