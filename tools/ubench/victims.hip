// Synthetic victims for tools/rsnt_beside_fft.py: which instruction class of a bystander kernel gets wrong results while the bf16 matrix
// kernel runs on the same device?  Each kernel is a long, register-only (or LDS-only) deterministic chain per lane that ends in one word
// per lane in memory; nothing depends on timing or on other workgroups.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/ubench/victims.hip -o tools/ab/libvictims.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v2 __attribute__((ext_vector_type(2)));

// 0: v_pk_fma_f32 chains (8 independent accumulators)      1: v_fma_f32 chains (16 independent accumulators)
// 2: LDS transposes (ds_write_b64 / barrier / ds_read_b64)  3: v_pk_mul_f32 + v_pk_add_f32      4: 64-bit integer mads
template <int KIND>
__global__ void __launch_bounds__(256) victimKernel(uint32_t iters, uint32_t *out)
{
    __shared__ v2 tile[256 * 9];
    const uint32_t tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
    const float seed = 1.0f + float(gid % 977u) * (1.0f / 1024.0f);
    uint32_t word = 0;
    if constexpr (KIND == 0 || KIND == 3) {
        v2 a[8], m = {0.999f + seed * 1e-4f, 1.001f - seed * 1e-4f}, c = {seed * 1e-3f, -seed * 1e-3f};
        for (int k = 0; k < 8; ++k) a[k] = v2{seed + k, seed - k};
        for (uint32_t i = 0; i < iters; ++i)
            for (int k = 0; k < 8; ++k) {
                if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                else { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c)); }
            }
        for (int k = 0; k < 8; ++k) word ^= __float_as_uint(a[k].x) * 31u + __float_as_uint(a[k].y);
    } else if constexpr (KIND == 1) {
        float a[16];
        const float m = 0.999f + seed * 1e-4f, c = seed * 1e-3f;
        for (int k = 0; k < 16; ++k) a[k] = seed + k;
        for (uint32_t i = 0; i < iters; ++i)
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
        for (int k = 0; k < 16; ++k) word ^= __float_as_uint(a[k]) * 31u + k;
    } else if constexpr (KIND == 2) {
        v2 a[8];
        for (int k = 0; k < 8; ++k) a[k] = v2{seed + k, seed - k};
        for (uint32_t i = 0; i < iters / 4; ++i) {
            for (int k = 0; k < 8; ++k) tile[(tid * 9 + k) % (256 * 9)] = a[k];
            __syncthreads();
            for (int k = 0; k < 8; ++k) a[k] = tile[(((tid + 37 * (k + 1)) & 255) * 9 + ((k + i) & 7)) % (256 * 9)];
            __syncthreads();
        }
        for (int k = 0; k < 8; ++k) word ^= __float_as_uint(a[k].x) * 31u + __float_as_uint(a[k].y);
    } else {
        unsigned long long a[8];
        for (int k = 0; k < 8; ++k) a[k] = gid * 2654435761ull + k;
        for (uint32_t i = 0; i < iters; ++i)
            for (int k = 0; k < 8; ++k) a[k] = a[k] * 6364136223846793005ull + 1442695040888963407ull;
        for (int k = 0; k < 8; ++k) word ^= uint32_t(a[k] >> 17);
    }
    out[gid] = word;
}

// 5 / 6: global loads only -- every lane XORs `iters` 16-byte words of a constant buffer (5: 1 MB, served by the caches; 6: 256 MB, from HBM)
__global__ void __launch_bounds__(256) victimLoadKernel(const uint4 *buf, uint32_t words, uint32_t iters, uint32_t *out)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    uint32_t at = (gid * 97u) % words;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint4 v = buf[at];
        acc.x ^= v.x; acc.y ^= v.y * 3u; acc.z ^= v.z * 5u; acc.w ^= v.w * 7u;
        at += 4099u; if (at >= words) at -= words;
    }
    out[gid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}
extern "C" int victim_load(const void *buf, uint32_t words, uint32_t blocks, uint32_t iters, uint32_t *d_out, void *stream)
{
    hipLaunchKernelGGL(victimLoadKernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint4 *>(buf), words, iters, d_out);
    return int(hipGetLastError());
}

extern "C" int victim_run(int kind, uint32_t blocks, uint32_t iters, uint32_t *d_out, void *stream)
{
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (kind) {
    case 0: hipLaunchKernelGGL(victimKernel<0>, dim3(blocks), dim3(256), 0, s, iters, d_out); break;
    case 1: hipLaunchKernelGGL(victimKernel<1>, dim3(blocks), dim3(256), 0, s, iters, d_out); break;
    case 2: hipLaunchKernelGGL(victimKernel<2>, dim3(blocks), dim3(256), 0, s, iters, d_out); break;
    case 3: hipLaunchKernelGGL(victimKernel<3>, dim3(blocks), dim3(256), 0, s, iters, d_out); break;
    default: hipLaunchKernelGGL(victimKernel<4>, dim3(blocks), dim3(256), 0, s, iters, d_out); break;
    }
    return int(hipGetLastError());
}
