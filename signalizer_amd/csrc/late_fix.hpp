// late_fix.hpp -- the pixels of a channel-split frame that need BOTH channels' spectra (spectrum_real.hip), as a function any kernel
// behind the channel workgroups' launch can apply: realLateKernel (its own launch) or K_B's fused kernel while it loads the magnitudes
// (spectrum_post.hip; then the step stays at two launches).
//
// csf[N/2] = |X_L[M] + i X_R[M]| / 2 (TransformDSP.inl:863) is the LAST offset of the arg-max scan of either side's top pixels
// (fixFrom[side] .. P) and is compared with a strict > (:957-979): it wins exactly when its square exceeds the winning square the
// channel's workgroup left in nyBest; the pixel then shows csf[N/2] itself, otherwise the value that workgroup wrote.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgz {

// mapAndTransformDFTFilters: magnitude = sqrt(re*re + im*im), im == 0 (TransformDSP.inl:1331,:1365).
// In binary floating point with round-to-nearest, sqrt(fl(x^2)) == |x| whenever x^2 neither underflows nor overflows (the square
// keeps |x| to half an ulp of the square, i.e. a quarter ulp of |x| after the root): the correctly rounded root -- a ~20-instruction
// sequence on this chip -- is needed outside [2^-62, 2^63] only.  tests/test_gpu_spectrum.py checks the identity over every float.
__device__ __forceinline__ float finishMagnitude(float val)
{
#pragma clang fp contract(off)
    const float a = __builtin_fabsf(val);
    if (a >= 0x1p-62f && a <= 0x1p63f) return a;
    const float sq = val * val + 0.f;
    return __builtin_sqrtf(sq);                                        // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
}

struct LateFix {
    __host__ __device__ uint32_t fixFrom(int side) const { return side ? fixFrom1 : fixFrom0; }
    const float *ny;          // [task][2]: the channels' Nyquist bins X_L[M], X_R[M] (bins injected: [task][0] is csf[N/2] itself)
    const float *nyBest;      // [task][2][64]: winning squares of the pixels fixFrom[side] + j
    uint32_t fixFrom0, fixFrom1;   // per side (two scalars, not an array: a dynamically indexed member would drag the kernel's whole argument
                                   // struct into addressable memory -- measured +8 us on every K_B kernel)
    uint32_t P;
    float invSize;
    uint32_t nyIsBin;         // sgz_stage_map_from_bins: ny[task][0] holds csf[N/2]
};

// csf[N/2] of a task, as the reference computes it from the packed bin (:863); nyRe / nyIm = lf.ny[2 task], lf.ny[2 task + 1]
__device__ __forceinline__ float lateNyquistValue(const LateFix &lf, float nyRe, float nyIm)
{
#pragma clang fp contract(off)
    return lf.nyIsBin ? nyRe : 0.5f * __builtin_amdgcn_sqrtf(nyRe * nyRe + nyIm * nyIm);
}
__device__ __forceinline__ float lateNyquistBin(const LateFix &lf, long task)
{
    return lateNyquistValue(lf, lf.ny[2 * task], lf.ny[2 * task + 1]);      // the left channel's is the real part
}
// the winning square the channel's workgroup left for pixel x >= fixFrom[side] of (task, side); +inf (csf[N/2] cannot win) outside the 64 slots
__device__ __forceinline__ float lateBestSquare(const LateFix &lf, long task, int side, uint32_t x)
{
    const uint32_t j = x - (side ? lf.fixFrom1 : lf.fixFrom0);
    return j < 64u ? lf.nyBest[size_t(2 * task + side) * 64 + j] : __builtin_inff();
}
// the pixel: `own` is what the channel's workgroup wrote, vM = csf[N/2], best = lateBestSquare()
__device__ __forceinline__ float lateNyquistPixel(const LateFix &lf, float vM, float best, float own)
{
#pragma clang fp contract(off)
    const float sqM = vM * vM + 0.f;                                    // Math::square(csf[offset]) with imag == 0
    return sqM > best ? finishMagnitude(lf.invSize * vM) : own;
}

}  // namespace sgz
