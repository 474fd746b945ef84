// fft_common.hpp -- device helpers shared by the fused FFT kernels (spectrum_fft.hip, spectrum_half.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgz {

// ---- compile-time twiddles W_32^j = cos(2 pi j/32) - i sin(2 pi j/32), j = 0..16 ---------------------
__host__ __device__ constexpr float cos32(int j)
{
    constexpr float v[17] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                             0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                             -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                             -0.92387953251128675613f, -0.98078528040323044913f, -1.0f};
    return v[j];
}
__host__ __device__ constexpr float sin32(int j) { return j <= 8 ? cos32(8 - j) : cos32(j - 8); }

__host__ __device__ constexpr int brev(int x, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
    return r;
}

// v2: two independent problems side by side (packed fp32 VALU ops); used by tools/ubench/dif.hip to compare
// scalar / packed / more-waves variants of the butterflies ("what bounds K_A").
typedef float v2 __attribute__((ext_vector_type(2)));

// In-register radix-2 DIF over LEN elements starting at BASE; result is in bit-reversed order.  V = float or v2.
template <typename V, int R, int LEN, int BASE>
__device__ __forceinline__ void dif(V (&re)[R], V (&im)[R])
{
    constexpr int H = LEN / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int a = BASE + i, b = BASE + i + H;
        const V ar = re[a], ai = im[a], br = re[b], bi = im[b];
        re[a] = ar + br;
        im[a] = ai + bi;
        const V dr = ar - br, di = ai - bi;
        const int j = i * (32 / LEN);
        if (j == 0) { re[b] = dr; im[b] = di; }
        else if (j == 8) { re[b] = di; im[b] = -dr; }
        else {
            const float c = cos32(j), s = sin32(j);
            re[b] = dr * c + di * s;
            im[b] = di * c - dr * s;
        }
    }
    if constexpr (LEN > 2) {
        dif<V, R, H, BASE>(re, im);
        dif<V, R, H, BASE + H>(re, im);
    }
}

// Buffer-resource (SRSRC) loads: one wave-uniform descriptor + a 32-bit per-lane offset + a scalar offset,
// so the 3R strided loads of a thread need no 64-bit address VGPRs, and reads past `bytes` return 0
// (that is the zero padding of prepareTransform, TransformDSP.inl:220-223, for W < N).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t makeRsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, int(bytes), 0x00020000);
}
__device__ __forceinline__ float bufLoad(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// NOTE: __builtin_amdgcn_raw_buffer_load_b64/_b128 are mis-lowered to a single buffer_load_dword by this
// ROCm 7.2 hipcc (verified in the ISA), so a complex twiddle is fetched as two dword loads.
__device__ __forceinline__ float2 bufLoad2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const float x = bufLoad(r, voff, soff);
    const float y = bufLoad(r, voff + 4, soff);
    return make_float2(x, y);
}

// An opaque copy of a per-thread constant: address arithmetic derived from it cannot be hoisted out of the frame
// loop (where it would pin VGPRs for the whole iteration and spill); it is recomputed where it is used instead.
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}


// Factorised twiddles: W^{x q} for q = 4a + b is B_a * A_b with A_b = W^{x b} (b = 1..3) and B_a = W^{x 4a}
// (a = 1..R/4-1), so a thread fetches 3 + R/4 - 1 complex values instead of R - 1 (10 instead of 31 at R = 32)
// and spends 4 VALU ops per product.  tw: table rows [A_1, A_2, A_3, B_1, .., B_{R/4-1}], row stride `rowBytes`.
template <int LR>
struct TwFactors {
    static constexpr int R = 1 << LR;
    static constexpr int NB = R / 4 - 1;
    float2 a[3];
    float2 b[NB];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int voff, int rowBytes)
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = bufLoad2(rs, voff, i * rowBytes);
#pragma unroll
        for (int i = 0; i < NB; ++i) b[i] = bufLoad2(rs, voff, (3 + i) * rowBytes);
    }
    // multiply the DIF output (bit-reversed order) by W^{x q}, q = 1..R-1
    __device__ __forceinline__ void apply(float (&re)[R], float (&im)[R]) const
    {
#pragma unroll
        for (int q = 1; q < R; ++q) {
            const int qa = q >> 2, qb = q & 3;
            float wx, wy;
            if (qa == 0) { wx = a[qb - 1].x; wy = a[qb - 1].y; }
            else if (qb == 0) { wx = b[qa - 1].x; wy = b[qa - 1].y; }
            else {
                wx = b[qa - 1].x * a[qb - 1].x - b[qa - 1].y * a[qb - 1].y;
                wy = b[qa - 1].x * a[qb - 1].y + b[qa - 1].y * a[qb - 1].x;
            }
            const int i = brev(q, LR);
            const float x = re[i], y = im[i];
            re[i] = x * wx - y * wy;
            im[i] = x * wy + y * wx;
        }
    }
};

}  // namespace sgz
