// LDS instruction cost on gfx950: clocks per wave-instruction per CU (whole-workgroup timing: 16 waves, one workgroup),
// full exec vs only lanes 0..31 active.  Conflict-free addresses (consecutive lanes -> consecutive elements).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP, bool HALF>
__global__ void __launch_bounds__(1024) k(long long *clk, float *out)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int width = (OP % 3 == 0) ? 4 : (OP % 3 == 1 ? 8 : 16);
    unsigned addr = unsigned(tid) * width;
    f4 v = {float(tid), 1.f, 2.f, 3.f};
    f2 v2 = {float(tid), 1.f};
    for (int i = tid; i < 16384; i += 1024) lds[i] = float(i);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (!HALF || lane < 32) {
#pragma unroll 1
        for (int it = 0; it < 64; ++it) {
            if (OP == 0) { REP16(asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v.x) : "memory");) }
            if (OP == 1) { REP16(asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v2) : "memory");) }
            if (OP == 2) { REP16(asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");) }
            if (OP == 3) { REP16(asm volatile("ds_read_b32 %0, %1" : "=v"(v.x) : "v"(addr) : "memory");) }
            if (OP == 4) { f2 d; REP16(asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); v.y += d.x; }
            if (OP == 5) { f4 d; REP16(asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");) asm volatile("s_waitcnt lgkmcnt(0)"); v.y += d.z; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { clk[2 * (tid >> 6)] = t0; clk[2 * (tid >> 6) + 1] = t1; }
    out[tid] = v.x + v.y;
}
template <int OP, bool HALF>
static void run(const char *name, long long *d_clk, float *d_out)
{
    std::vector<long long> h(32);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((k<OP, HALF>), dim3(1), dim3(1024), 65536, 0, d_clk, d_out);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_clk, 32 * sizeof(long long), hipMemcpyDeviceToHost);
        long long lo = h[0], hi = h[1];
        for (int w = 0; w < 16; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
        best = std::min(best, double(hi - lo));
    }
    printf("%-14s %-10s %7.2f clk / wave-instruction / CU\n", name, HALF ? "lanes<32" : "full", best / (64.0 * 16 * 16));
}
int main()
{
    long long *d_clk; float *d_out;
    hipMalloc(&d_clk, 32 * sizeof(long long)); hipMalloc(&d_out, 1024 * sizeof(float));
    run<0, false>("ds_write_b32", d_clk, d_out);  run<0, true>("ds_write_b32", d_clk, d_out);
    run<1, false>("ds_write_b64", d_clk, d_out);  run<1, true>("ds_write_b64", d_clk, d_out);
    run<2, false>("ds_write_b128", d_clk, d_out); run<2, true>("ds_write_b128", d_clk, d_out);
    run<3, false>("ds_read_b32", d_clk, d_out);   run<3, true>("ds_read_b32", d_clk, d_out);
    run<4, false>("ds_read_b64", d_clk, d_out);   run<4, true>("ds_read_b64", d_clk, d_out);
    run<5, false>("ds_read_b128", d_clk, d_out);  run<5, true>("ds_read_b128", d_clk, d_out);
    return 0;
}
