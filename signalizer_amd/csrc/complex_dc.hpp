// complex_dc.hpp -- the csf entries that mapToLinearSpace leaves COMPLEX, and the pixels that reach them (shared by the fused,
// halves and generic map paths).
//
// After the |.| pass of the reference most of csf holds magnitudes, but not all of it:
//   Complex mode:                 csf[0] = Z[0] / 2 stays complex                          (TransformDSP.inl:993, :999-1002)
//   Left / Right / Merge / Side:  only i < N/2 become magnitudes (:553-560); csf[N/2] = Z[N/2] / 2 and csf[k] = Z[k],
//                                 N/2 < k < N, stay raw -- and a filter window below bin 0 wraps onto csf[N-1], csf[N-2], ..
//                                 (periodic indexing over the N + 1 entries), one near Nyquist reaches csf[N/2 ..].
// A pixel whose taps or arg-max run touch such an entry is a complex sum (linearFilter / lanczosFilter on std::complex,
// :588 / :620 / :1020 / :1035) or compares re^2 + im^2 (Math::square of a complex) and may select a complex value;
// mapAndTransformDFTFilters then takes sqrt(re^2 + im^2) (:1331).  The map kernels work on magnitudes only; plan.cpp lists these
// pixels (dcPixels) and they are redone here, in the oracle's operation order (oracle/spectrum.c), from the magnitudes plus the
// kSpecBins complex entries a pixel can reach.
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"

namespace sgz {

// slot of csf[k] among the complex entries kept per task, or -1 when csf[k] is a magnitude:
//   slots 0..7  <-> k = N-8 .. N-1   (mono modes: the wrap below bin 0);   Complex mode: slot 0 <-> k = 0
//   slots 8..15 <-> k = N/2 .. N/2+7 (mono modes: around Nyquist)
constexpr int kSpecBins = 16;
__host__ __device__ inline int specSlot(uint32_t mode, int N, int k)
{
    if (mode == SGZ_CH_COMPLEX) return k == 0 ? 0 : -1;
    if (mode == SGZ_CH_LEFT || mode == SGZ_CH_RIGHT || mode == SGZ_CH_MERGE || mode == SGZ_CH_SIDE) {
        if (k >= N - 8 && k < N) return k - (N - 8);
        if (k >= N / 2 && k < N / 2 + 8) return 8 + k - N / 2;
    }
    return -1;
}
// the factor the reference applies to that entry before the |.| pass (:553-554, :993)
__host__ __device__ inline float specScale(uint32_t mode, int N, int k)
{
    return (mode == SGZ_CH_COMPLEX ? k == 0 : k == N / 2) ? 0.5f : 1.0f;
}

// One listed pixel.  fetch(k): the magnitude csf[k] of an ordinary entry; spec(slot): the complex entry of that slot.
template <typename Fetch, typename Spec>
__device__ __forceinline__ float complexDcPixel(const PixelRec rec, const float *weights, float invSize, int N, uint32_t mode,
                                                Fetch fetch, Spec spec)
{
#pragma clang fp contract(off)
    auto entry = [&](int k) {
        const int s = specSlot(mode, N, k);
        return s >= 0 ? spec(s) : make_float2(fetch(k), 0.f);
    };
    float re, im;
    if ((rec.kind & 1) == 0) {
        float ar = 0.f, ai = 0.f;
        int k = rec.a;
        for (int i = 0; i < rec.b; ++i) {
            const float w = weights[rec.c + i];
            const float2 v = entry(k);
            const float pr = v.x * w, pi = v.y * w;
            ar = ar + pr;
            ai = ai + pi;
            k = (k == N) ? 0 : k + 1;
        }
        re = invSize * ar;
        im = invSize * ai;
    } else {
        float best = 0.f;
        int arg = rec.c;
        for (int i = 0; i < rec.b; ++i) {
            const int k = rec.a + i;
            const float2 v = entry(k);
            const float a = v.x * v.x, b = v.y * v.y;
            const float sq = a + b;
            if (sq > best) { best = sq; arg = k; }
        }
        const float2 v = entry(arg);
        re = invSize * v.x;
        im = invSize * v.y;
    }
    const float a = re * re, b = im * im;
    return __builtin_sqrtf(a + b);
}

}  // namespace sgz
