// complex_dc.hpp -- SpectrumChannels::Complex: the pixels that touch csf[0] (shared by the fused and the generic map kernels).
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"

namespace sgz {

// One pixel of SpectrumChannels::Complex whose filter taps or arg-max run include bin 0.  The reference halves csf[0] but
// leaves it complex (TransformDSP.inl:993; every other bin becomes its magnitude, :999-1002), so the pixel is a complex
// sum (linearFilter / lanczosFilter on std::complex, :1020 / :1035) or, in an arg-max run, competes with re^2 + im^2
// (Math::square of a complex) and may be selected as a complex value; mapAndTransformDFTFilters then takes
// sqrt(re^2 + im^2) (:1331).  fetch(k): csf[k] for k != 0.  Same operation order as the oracle (oracle/spectrum.c).
template <typename Fetch>
__device__ __forceinline__ float complexDcPixel(const PixelRec rec, const float *weights, float invSize, int N, float re0, float im0,
                                                Fetch fetch)
{
#pragma clang fp contract(off)
    float re, im;
    if ((rec.kind & 1) == 0) {
        float ar = 0.f, ai = 0.f;
        int k = rec.a;
        for (int i = 0; i < rec.b; ++i) {
            const float w = weights[rec.c + i];
            const float vr = k == 0 ? re0 : fetch(k), vi = k == 0 ? im0 : 0.f;
            const float pr = vr * w, pi = vi * w;
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
            float sq;
            if (k == 0) { const float a = re0 * re0, b = im0 * im0; sq = a + b; }
            else { const float m = fetch(k); sq = m * m + 0.f; }
            if (sq > best) { best = sq; arg = k; }
        }
        re = invSize * (arg == 0 ? re0 : fetch(arg));
        im = invSize * (arg == 0 ? im0 : 0.f);
    }
    const float a = re * re, b = im * im;
    return __builtin_sqrtf(a + b);
}

}  // namespace sgz
