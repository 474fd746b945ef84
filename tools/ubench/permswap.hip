// does the two-swap exchange give every lane its partner's (lane ^ 32) pair?  hipcc --offload-arch=gfx950 -O3 permswap.hip -o permswap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, float *out)
{
    const int t = threadIdx.x;
    v2 c[4];
    for (int i = 0; i < 4; ++i) c[i] = v2{in[8 * t + 2 * i], in[8 * t + 2 * i + 1]};
    // (the builtin form -- r1 = __builtin_amdgcn_permlane32_swap(x, y); r2 = ...swap(r1[1], r1[0]) -- is miscompiled by ROCm 7.2's hipcc:
    // it overwrites x with y in front of the first swap; compile with -DBUILTIN to see it)
    {
        float x0 = c[0].x, y0 = c[0].y, x1 = c[1].x, y1 = c[1].y, x2 = c[2].x, y2 = c[2].y, x3 = c[3].x, y3 = c[3].y;
        asm volatile("s_nop 1\n\t"
                     "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                     "v_permlane32_swap_b32 %1, %0\n\tv_permlane32_swap_b32 %3, %2\n\tv_permlane32_swap_b32 %5, %4\n\tv_permlane32_swap_b32 %7, %6"
                     : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(x3), "+v"(y3));
        c[0] = v2{y0, x0}; c[1] = v2{y1, x1}; c[2] = v2{y2, x2}; c[3] = v2{y3, x3};      // the pair comes out crossed
    }
    for (int i = 0; i < 4; ++i) { out[8 * t + 2 * i] = c[i].x; out[8 * t + 2 * i + 1] = c[i].y; }
}
int main()
{
    float h[512], o[512], *d, *e;
    for (int i = 0; i < 512; ++i) h[i] = float(i);
    hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) for (int j = 0; j < 8; ++j) bad += o[8 * t + j] != h[8 * (t ^ 32) + j];
    printf("bad %d   lane0: %g %g %g %g (want 256 257 258 259)\n", bad, o[0], o[1], o[2], o[3]);
    return 0;
}
