// fft_scalar.hpp -- in-register radix-2 butterflies in decimation-in-time form, p = a + w b, q = 2 a - p, which costs three
// fused operations per general twiddle instead of the four (subtract, then a complex multiply) of decimation in frequency.
//
// What the instructions cost on this chip with the register traffic of a real butterfly network (tools/ubench/valu4.hip, banks.hip,
// dit.hip; NOTES.md "VALU rates"): a wave64 plain fp32 v_add / v_mul / v_fmac / v_fmamk occupies its SIMD for ~2.8-3.1 clocks
// (2.2 only in a dependent chain on one register), a three-source v_fma 3.1-3.7 (5.5 when two sources share a VGPR bank), a
// v_pk_*_f32 4.8-5.5: per flop the packed form is ~10 % cheaper and it halves the issue slots.  Whole 32-point transforms on 8 waves
// of one CU: difPacked 1153 clocks, the scalar DIT below 1427, the packed DIT (ditPacked, at the end of this file) 998 -- the kernels
// use ditPacked; the scalar form stays for the microbenchmark and as the readable statement of the arithmetic.
//     p.re = fma(b.im, s, fma(b.re, c, a.re))    p.im = fma(b.re, -s, fma(b.im, c, a.im))    q = fma(2, a, -p)
//
// Register convention (the same as difPacked in fft_common.hpp, so the kernels' exchanges do not change): input x[j] in c[BASE + j],
// output X[k] in c[BASE + brev(k)].  DIT wants its input in bit-reversed order and delivers natural order; here the array is simply
// read through brev() (compile-time indices of a fully unrolled loop: the permutation is free).
#pragma once
#include "fft_common.hpp"

namespace sgz {

// c * w for a per-lane w = (w.x, w.y): 4 plain operations
__device__ __forceinline__ v2 cmulScalar(v2 c, float2 w)
{
    const float t0 = c.x * w.x, t1 = c.x * w.y;
    return v2{__builtin_fmaf(-c.y, w.y, t0), __builtin_fmaf(c.y, w.x, t1)};
}
__device__ __forceinline__ float2 cmulScalar(float2 c, float2 w)
{
    const float t0 = c.x * w.x, t1 = c.x * w.y;
    return float2{__builtin_fmaf(-c.y, w.y, t0), __builtin_fmaf(c.y, w.x, t1)};
}
// w * w: 3 operations
__device__ __forceinline__ float2 csqScalar(float2 w)
{
    const float d = w.x * w.x;
    return float2{__builtin_fmaf(-w.y, w.y, d), (w.x + w.x) * w.y};
}

// one DIT stage: butterflies of span 2^S / 2 on the logical array y[i] = c[BASE + brev(i)], twiddles W_{2^S}^j = W_32^{j 32 / 2^S}
template <int LR, int S, int BASE, int NREG>
__device__ __forceinline__ void ditStage(v2 (&c)[NREG])
{
    constexpr int n = 1 << LR, m = 1 << S, h = m / 2;
#pragma unroll
    for (int k0 = 0; k0 < n; k0 += m) {
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const int ia = BASE + brev(k0 + j, LR), ib = BASE + brev(k0 + j + h, LR);
            const v2 a = c[ia], b = c[ib];
            const int tw = j * (32 / m);                                // W_32^tw, tw in [0, 16)
            if (tw == 0) { c[ia] = a + b; c[ib] = a - b; }
            else if (tw == 8) { c[ia] = v2{a.x + b.y, a.y - b.x}; c[ib] = v2{a.x - b.y, a.y + b.x}; }      // w b = -i b
            else if (tw < 8) {
                const float cs = cos32(tw), sn = sin32(tw);             // w b = (cs b.re + sn b.im, cs b.im - sn b.re)
                const float pr = __builtin_fmaf(b.y, sn, __builtin_fmaf(b.x, cs, a.x));
                const float pi = __builtin_fmaf(b.x, -sn, __builtin_fmaf(b.y, cs, a.y));
                c[ia] = v2{pr, pi};
                c[ib] = v2{__builtin_fmaf(2.f, a.x, -pr), __builtin_fmaf(2.f, a.y, -pi)};
            } else {
                const float cs = cos32(tw - 8), sn = sin32(tw - 8);     // w = -i w', w' b = (tr, ti):  w b = (ti, -tr)
                const float pr = __builtin_fmaf(b.x, -sn, __builtin_fmaf(b.y, cs, a.x));
                const float pi = __builtin_fmaf(b.y, -sn, __builtin_fmaf(b.x, -cs, a.y));
                c[ia] = v2{pr, pi};
                c[ib] = v2{__builtin_fmaf(2.f, a.x, -pr), __builtin_fmaf(2.f, a.y, -pi)};
            }
        }
    }
}

// 2^LR-point transform of c[BASE .. BASE + 2^LR): x[j] in c[BASE + j] -> X[k] in c[BASE + brev(k, LR)].  FIRST: the first stage to run
// (a caller that has already combined the pairs (j, j + n/2) -- the window-fused load of spectrum_real.hip -- starts at 2)
template <int LR, int BASE, int NREG, int FIRST = 1>
__device__ __forceinline__ void ditScalar(v2 (&c)[NREG])
{
    if constexpr (FIRST <= 1) ditStage<LR, 1, BASE>(c);
    if constexpr (FIRST <= 2 && LR >= 2) ditStage<LR, 2, BASE>(c);
    if constexpr (LR >= 3) ditStage<LR, 3, BASE>(c);
    if constexpr (LR >= 4) ditStage<LR, 4, BASE>(c);
    if constexpr (LR >= 5) ditStage<LR, 5, BASE>(c);
}

// ---- the same decimation-in-time butterflies on packed (re, im) pairs: 3 packed operations for a general twiddle instead of the 4 of
// the decimation-in-frequency form in fft_common.hpp (difPacked), 2 for w = 1 and w = -i:
//     t = a + c b                          v_pk_fma_f32  b, (c, s), a      op_sel_hi:[1,0,1]                       (both halves times c)
//     p = t + s (b.im, -b.re)              v_pk_fma_f32  b, (c, s), t      op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]
//     q = 2 a - p                          v_pk_fma_f32  a, 2.0, p         neg_lo / neg_hi on p
// and for w = -i w':  t = a + c (b.im, -b.re),  p = t - s b.  Measured with realistic register traffic (tools/ubench/valu4.hip,
// banks.hip) a packed operation costs ~4.8-5.5 clocks against ~2.9-3.1 for each of the two plain ones it replaces, and half the
// issue slots -- which is what a workgroup alone on its CU is short of.  A 32-point transform: 46 x 2 + 34 x 3 = 194 packed
// operations (difPacked: 228).
// twiddle constants travel as an SGPR pair k = (cos, sin)
__device__ __forceinline__ void bflyPackedLow(v2 &a, v2 &b, v2 k)            // w = k.x - i k.y,  0 < angle < pi/2
{
    v2 t, p, q;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(b), "s"(k), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(p) : "v"(b), "s"(k), "v"(t));
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(p));
    a = p; b = q;
}
__device__ __forceinline__ void bflyPackedHigh(v2 &a, v2 &b, v2 k)           // w = -i (k.x - i k.y)
{
    v2 t, p, q;
    // t = (a.x + c b.y, a.y - c b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(t) : "v"(b), "s"(k), "v"(a));
    // p = (t.x - s b.x, t.y - s b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(p) : "v"(b), "s"(k), "v"(t));
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(a), "v"(p));
    a = p; b = q;
}
// p = a - i b, q = a + i b   (w = -i)
__device__ __forceinline__ void bflyPackedRot(v2 &a, v2 &b)
{
    v2 p, q;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(p) : "v"(a), "v"(b));      // (a.x + b.y, a.y - b.x)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(q) : "v"(a), "v"(b));      // (a.x - b.y, a.y + b.x)
    a = p; b = q;
}

// A first-level (twiddle-free) butterfly with the twiddles its two inputs still owe from the pass before -- p = ta a + tb b, q = ta a - tb b
// -- as u = ta a (2 operations; none where ta = 1), p = u + tb b (2), q = 2 u - p (1): five packed operations where "multiply both,
// then add and subtract" takes six.  ta, tb per lane (vector registers).
template <bool UNIT_A>
__device__ __forceinline__ void bflyPackedFusedTw(v2 &a, v2 &b, v2 ta, v2 tb)
{
    v2 u = a, t, p, q;
    if constexpr (!UNIT_A) {
        v2 m;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(m) : "v"(a), "v"(ta));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(u) : "v"(a), "v"(ta), "v"(m));
    }
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(t) : "v"(b), "v"(tb), "v"(u));                                  // u + b tb.re
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(p) : "v"(b), "v"(tb), "v"(t));                     // + (-b.im, b.re) tb.im
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(q) : "v"(u), "v"(p));
    a = p; b = q;
}

template <int LR, int S, int BASE, int NREG>
__device__ __forceinline__ void ditStagePacked(v2 (&c)[NREG])
{
    constexpr int n = 1 << LR, m = 1 << S, h = m / 2;
#pragma unroll
    for (int k0 = 0; k0 < n; k0 += m) {
#pragma unroll
        for (int j = 0; j < h; ++j) {
            const int ia = BASE + brev(k0 + j, LR), ib = BASE + brev(k0 + j + h, LR);
            const int tw = j * (32 / m);                                // W_32^tw, tw in [0, 16)
            if (tw == 0) { const v2 a = c[ia], b = c[ib]; c[ia] = a + b; c[ib] = a - b; }
            else if (tw == 8) bflyPackedRot(c[ia], c[ib]);
            else if (tw < 8) bflyPackedLow(c[ia], c[ib], v2{cos32(tw), sin32(tw)});
            else bflyPackedHigh(c[ia], c[ib], v2{cos32(tw - 8), sin32(tw - 8)});
        }
    }
}
// 2^LR-point transform of c[BASE .. BASE + 2^LR): x[j] in c[BASE + j] -> X[k] in c[BASE + brev(k, LR)] (difPacked's convention)
template <int LR, int BASE, int NREG, int FIRST = 1>
__device__ __forceinline__ void ditPacked(v2 (&c)[NREG])
{
    if constexpr (FIRST <= 1) ditStagePacked<LR, 1, BASE>(c);
    if constexpr (FIRST <= 2 && LR >= 2) ditStagePacked<LR, 2, BASE>(c);
    if constexpr (LR >= 3) ditStagePacked<LR, 3, BASE>(c);
    if constexpr (LR >= 4) ditStagePacked<LR, 4, BASE>(c);
    if constexpr (LR >= 5) ditStagePacked<LR, 5, BASE>(c);
}

}  // namespace sgz
