// per-CU streaming-read microbenchmark: how fast can ONE workgroup (8 or 16 waves) pull 384 KB that is L2-resident,
// with (a) the FFT kernel's access pattern (dword, stride 4096 B between a thread's loads) and (b) contiguous dwordx4?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(const float *src, float *out, long long *clk, int perThreadDwords, size_t wgStride)
{
    const float *p = src + size_t(blockIdx.x) * wgStride;
    const int tid = threadIdx.x;
    float acc = 0.f;
    long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {            // dword, thread-strided by THREADS (coalesced per instruction, 4*THREADS B between a thread's loads)
#pragma unroll 32
        for (int j = 0; j < perThreadDwords; ++j) acc += p[tid + THREADS * j];
    } else if (MODE == 1) {     // dwordx4 contiguous per instruction
        const float4 *q = reinterpret_cast<const float4 *>(p);
#pragma unroll 16
        for (int j = 0; j < perThreadDwords / 4; ++j) { float4 v = q[tid + THREADS * j]; acc += v.x + v.y + v.z + v.w; }
    } else if (MODE == 3) {     // raw buffer loads, soffset = j * 4 * THREADS (what the FFT kernel does), 3 arrays
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 128 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + 32768), 0, 128 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + 65536), 0, 128 * 1024, 0x00020000);
        float v[192];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            v[3 * j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, tid * 4, j * 4 * THREADS, 0));
            v[3 * j + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, tid * 4, j * 4 * THREADS, 0));
            v[3 * j + 2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, tid * 4, j * 4 * THREADS, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 192; ++j) acc += v[j];
    } else if (MODE == 4) {     // exactly K_A's pattern: 1024 threads, L / R / window, 32 dwords each, 4 KB between a thread's loads
        extern __shared__ float dyn[];
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 128 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + 32768), 0, 128 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + 65536), 0, 128 * 1024, 0x00020000);
        float v[96];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            v[3 * j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, tid * 4, j * 4096, 0));
            v[3 * j + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, tid * 4, j * 4096, 0));
            v[3 * j + 2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, tid * 4, j * 4096, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 96; ++j) acc += v[j];
        if (perThreadDwords < 0) dyn[tid] = acc;
    } else if (MODE == 5 || MODE == 6 || MODE == 7) {     // K_A's three arrays, but at most 32 (5), 64 (6) or 16 (7) loads in flight per wave
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 128 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + 32768), 0, 128 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p + 65536), 0, 128 * 1024, 0x00020000);
        constexpr int B = MODE == 5 ? 32 : (MODE == 6 ? 64 : 16);      // loads per batch
        float v[96];
#pragma unroll
        for (int b0 = 0; b0 < 96; b0 += B) {
#pragma unroll
            for (int i = b0; i < b0 + B && i < 96; ++i) {
                const int arr = i / 32, j = i % 32;
                v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(arr == 0 ? r0 : (arr == 1 ? r1 : r2), tid * 4, j * 4096, 0));
            }
            __builtin_amdgcn_s_waitcnt(0x0070 | 0x3f00);   // vmcnt(0) only (gfx9 encoding: vmcnt lo [3:0], hi [15:14])
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 96; ++j) acc += v[j];
    } else if (MODE == 9) {     // buffer loads, ONE contiguous 384 KB array, 96 in flight
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 384 * 1024, 0x00020000);
        float v[96];
#pragma unroll
        for (int j = 0; j < 96; ++j) v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, tid * 4, j * 4096, 0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 96; ++j) acc += v[j];
    } else if (MODE == 10) {    // global loads, three arrays interleaved per j, 96 in flight
        float v[96];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            v[3 * j] = p[tid + 1024 * j];
            v[3 * j + 1] = p[32768 + tid + 1024 * j];
            v[3 * j + 2] = p[65536 + tid + 1024 * j];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 96; ++j) acc += v[j];
    } else if (MODE == 11) {    // global loads, one array, 96 in flight, consumed after all issued
        float v[96];
#pragma unroll
        for (int j = 0; j < 96; ++j) v[j] = p[tid + 1024 * j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 96; ++j) acc += v[j];
    } else if (MODE == 8) {     // one array per phase, global loads, 32 in flight
        float v[96];
#pragma unroll
        for (int arr = 0; arr < 3; ++arr) {
#pragma unroll
            for (int j = 0; j < 32; ++j) v[arr * 32 + j] = p[arr * 32768 + tid + 1024 * j];
            __builtin_amdgcn_s_waitcnt(0x0070 | 0x3f00);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 96; ++j) acc += v[j];
    } else {                    // dwordx2
        const float2 *q = reinterpret_cast<const float2 *>(p);
#pragma unroll 32
        for (int j = 0; j < perThreadDwords / 2; ++j) { float2 v = q[tid + THREADS * j]; acc += v.x + v.y; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * THREADS + tid] = acc;
    // workgroup time = last wave's end - first wave's start (a wave-0-only clock hides the queueing of later waves)
    __shared__ long long s0[16], s1[16];
    if ((tid & 63) == 0) { s0[tid >> 6] = t0; s1[tid >> 6] = t1; }
    __syncthreads();
    if (tid == 0) {
        long long a = s0[0], b = s1[0];
        for (int w = 1; w < THREADS / 64; ++w) { a = s0[w] < a ? s0[w] : a; b = s1[w] > b ? s1[w] : b; }
        clk[blockIdx.x] = b - a;
    }
}

int main()
{
    const size_t bytesPerWg = 384 * 1024;
    const int wgs = 256;
    float *src, *out; long long *clk;
    CK(hipMalloc(&src, bytesPerWg * wgs)); CK(hipMalloc(&out, 4 * 1024 * wgs)); CK(hipMalloc(&clk, 8 * wgs));
    CK(hipMemset(src, 0, bytesPerWg * wgs));
    std::vector<long long> h(wgs);
    size_t ldsBytes = 0;
    auto run = [&](const char *name, auto kern, int threads, int nwg, size_t stride) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), ldsBytes, 0, src, out, clk, int(bytesPerWg / 4 / threads), stride);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), clk, 8 * nwg, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < nwg; ++i) avg += h[i]; avg /= nwg;
        printf("%-44s wgs=%3d  cycles=%8.0f  -> %6.1f B/clk/CU\n", name, nwg, avg, bytesPerWg / avg);
        return 0;
    };
    // same data for every WG (L2-hot after first rep) vs distinct data per WG
    for (int nwg : {1, 256}) {
        for (size_t stride : {size_t(0), bytesPerWg / 4}) {
            printf("--- nwg=%d %s\n", nwg, stride ? "distinct data per WG" : "shared data (L2 hot)");
            run("dword  512 thr", k<0, 512>, 512, nwg, stride);
            run("dword 1024 thr", k<0, 1024>, 1024, nwg, stride);
            run("dwordx2 512 thr", k<2, 512>, 512, nwg, stride);
            run("dwordx4 512 thr", k<1, 512>, 512, nwg, stride);
            run("dwordx4 1024 thr", k<1, 1024>, 1024, nwg, stride);
            run("buffer dword 512 thr 3 arrays, 192 in flight", k<3, 512>, 512, nwg, stride);
            run("K_A pattern 1024 thr, no LDS", k<4, 1024>, 1024, nwg, stride);
            run("K_A arrays, 32 in flight", k<5, 1024>, 1024, nwg, stride);
            run("K_A arrays, 64 in flight", k<6, 1024>, 1024, nwg, stride);
            run("K_A arrays, 16 in flight", k<7, 1024>, 1024, nwg, stride);
            run("K_A arrays, global loads, 32 in flight", k<8, 1024>, 1024, nwg, stride);
            run("buffer loads, one array, 96 in flight", k<9, 1024>, 1024, nwg, stride);
            run("global loads, 3 arrays interleaved, 96", k<10, 1024>, 1024, nwg, stride);
            run("global loads, one array, 96 in flight", k<11, 1024>, 1024, nwg, stride);
            ldsBytes = 150 * 1024;
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k<4, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            run("K_A pattern 1024 thr, 150 KB LDS", k<4, 1024>, 1024, nwg, stride);
            ldsBytes = 0;
        }
    }
    return 0;
}
