// decay_body.hpp -- the per-(frame, pixel) arithmetic of K_B (peak decay replay, dB map, colour blend), shared by the kernels
// of spectrum_post.hip.  Include it AFTER any code that wants fp contraction: everything below this line rounds exactly
// like the reference's scalar code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

#pragma clang fp contract(off)

namespace sgz {

constexpr int kMaxChunk = kDecayChunk;
constexpr int G = SGZ_NUM_GRAPHS;
constexpr int NC = SGZ_NUM_SPEC_COLOURS + 1;

// std::log(float) as glibc >= 2.27 computes it (sysdeps/ieee754/flt-32/e_logf.c, Szabolcs Nagy's algorithm: x = 2^k z,
// z in [0x3f330000, 2x), 16 sub-intervals with tabulated 1/c and log(c), a cubic in r = z/c - 1 evaluated in double, one
// rounding to float).  This is what the reference's `std::log(T)` resolves to on Linux and what the oracle calls, so the dB
// values -- and with them the truncated uint8 colours -- are bit-identical.  Checked exhaustively against libm's logf over every
// positive finite float, with and without fused multiply-adds in the double arithmetic (the x86-64 ifunc variants): no
// difference either way, the double result never sits that close to a float rounding boundary (tests/test_oracle_math.py).
// x > 0 (possibly subnormal or +inf); the caller has excluded zero, negatives and NaN.
__device__ static const double kLogfTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
// tab: the table's home -- constant memory, or a workgroup's copy in LDS (the fused K_B kernels: two table reads per dB value are
// dependent loads in front of the fp64 chain)
typedef const double (*LogfTable)[2];
__device__ __forceinline__ float glibcLogf(float x, LogfTable tab = kLogfTab)
{
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix == 0x7f800000u) return x;                        // log(inf) = inf
        ix = __float_as_uint(x * 0x1p23f);                      // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = int((tmp >> 19) & 15u);
    const int k = int(tmp) >> 23;                               // arithmetic shift
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = tab[i][0], logc = tab[i][1];
    const double z = double(__uint_as_float(iz));
    const double r = z * invc - 1.0;
    const double y0 = logc + double(k) * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return float(y);
}

__device__ __forceinline__ float dbMap(float slope, float st, const DeviceScalars &sc, LogfTable tab = kLogfTab)
{
    const float deltaX = slope * st * sc.minFracRecip;          // :1343 (left-to-right fp32)
    return deltaX > 0.f ? glibcLogf(deltaX, tab) * sc.deltaYRecip : sc.lowerClip;   // :1345, std::log(float) = logf
}
// the table into LDS (every thread of the workgroup calls; a barrier must follow before the first use)
__device__ __forceinline__ void stageLogfTable(double (*dst)[2])
{
    if (threadIdx.x < 32) reinterpret_cast<double *>(dst)[threadIdx.x] = reinterpret_cast<const double *>(kLogfTab)[threadIdx.x];
}

// renderSf + the additive blend of one pair's colour into the column buffer, SpectrumDSP.cpp:119-174
// the colour a pair contributes at `intensity` (false: nothing, intensity < 0), and its blend into the column buffer
__device__ __forceinline__ bool colourOf(float intensity, const float *sca, const DeviceScalars &sc, float (&colourv)[3])
{
    if (intensity < 0.f) return false;
    colourv[0] = sca[(NC - 1) * 3 + 0]; colourv[1] = sca[(NC - 1) * 3 + 1]; colourv[2] = sca[(NC - 1) * 3 + 2];
    if (intensity < 0.999f) {
        float accumulatedSum = 0.f;
        for (int c = 1; c < NC; ++c) {
            const float nextScale = sc.ratios[c];
            accumulatedSum += nextScale;
            if (accumulatedSum >= intensity) {
                const float mn = accumulatedSum - nextScale;
                const float mx = accumulatedSum;
                const float mix = (intensity - mn) / (mx - mn);
                const float imix = 1.f - mix;
                const float *ca = sca + (c - 1) * 3, *cbb = sca + c * 3;
                colourv[0] = ca[0] * imix + cbb[0] * mix;
                colourv[1] = ca[1] * imix + cbb[1] * mix;
                colourv[2] = ca[2] * imix + cbb[2] * mix;
                break;
            }
        }
    }
    return true;
}
__device__ __forceinline__ void screenBlend(float (&cb)[3], const float (&colourv)[3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) cb[c] += (1.f - cb[c]) * colourv[c];   // GL_ONE_MINUS_SRC_COLOR
}
__device__ __forceinline__ void blendColour(float (&cb)[3], float intensity, const float *sca, const DeviceScalars &sc)
{
    float colourv[3];
    if (colourOf(intensity, sca, sc, colourv)) screenBlend(cb, colourv);
}

__device__ __forceinline__ uchar4 toRgba8(const float (&cb)[3])
{
    uchar4 pxl;
    pxl.x = (unsigned char)(cb[0] * 255.f);                     // static_cast<uint8_t>(c * maxByte), SpectrumDSP.cpp:195-198
    pxl.y = (unsigned char)(cb[1] * 255.f);
    pxl.z = (unsigned char)(cb[2] * 255.f);
    pxl.w = 255;
    return pxl;
}

// dB map + colour blend + line / state outputs of ONE (frame, pixel): replays the chunk's recurrence up to frame
// f = f0 + t (<= 8 steps) on top of the chunk's carry-in states.  carryIn[m * carryStride] = exact state of
// (pair, side, graph) combination m = (pair * sides + side) * G + graph at the end of the previous chunk (unused for
// chunk 0, where the caller's carry-in state applies instead).
// colourTab: prm.colourTables or a copy in LDS; mag0: the magnitudes of (pair 0, side 0), already loaded, or null.
__device__ __forceinline__ void emitPixel(const DecayParams &prm, uint32_t chunk, int t, uint32_t pixel, bool allCombos,
                                          const float *carryIn, uint32_t carryStride, const float *colourTab, float slope,
                                          const float *mag0)
{
    const size_t perFrame = size_t(prm.C) * prm.sides * prm.P;
    const long f0 = long(chunk) * kMaxChunk, f = f0 + t;
    float cb[3] = {0.f, 0.f, 0.f};                              // colourBuffer, SpectrumDSP.cpp:170-174
    for (uint32_t pair = 0; pair < prm.C; ++pair) {
        const float *sca = colourTab + size_t(pair) * NC * 3;
        for (uint32_t side = 0; side < prm.sides; ++side) {
            if (!allCombos && side != 0) continue;              // only (side 0, graph 0) feeds the colour column
            const uint32_t ps = pair * prm.sides + side;
            float mag[kMaxChunk];
            if (ps == 0 && mag0) {
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i) mag[i] = mag0[i];
            } else {
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i)             // independent loads first
                    mag[i] = i <= t ? prm.mapped[size_t(f0 + i) * perFrame + size_t(ps) * prm.P + pixel] * prm.magScale : 0.f;
            }
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const bool colour = (side == 0 && k == 0 && prm.rgba);
                if (!colour && !allCombos) continue;
                const float pole = prm.sc.pole[k];
                float cr = chunk > 0 ? carryIn[(ps * G + k) * carryStride] : 0.f;
                float a = (chunk == 0 && prm.stateIn) ? prm.stateIn[((size_t(pair) * G + k) * prm.P + pixel) * 2 + side] : 0.f;
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i) {
                    if (i <= t) {
                        a = a * pole;                           // states[i] *= pole, TransformDSP.inl:1336,:1370
                        if (mag[i] > a) a = mag[i];             // :1338-1341
                        cr = cr * pole;
                    }
                }
                const float s = a > cr ? a : cr;
                if (prm.state && f == prm.frames - 1)
                    prm.state[((size_t(pair) * G + k) * prm.P + pixel) * 2 + side] = s;
                if (!colour && !prm.lines) continue;
                const float result = dbMap(slope, s, prm.sc);
                if (prm.lines) {
                    float *lr = prm.lines + (((size_t(f) * prm.C + pair) * G + k) * prm.P + pixel) * 2;
                    lr[side] = result;
                    if (prm.sides == 1) lr[1] = 0.f;            // results[i].phase = 0 in the one-channel modes (:1347)
                }
                if (colour) blendColour(cb, result, sca, prm.sc);
            }
        }
    }
    if (prm.rgba) reinterpret_cast<uchar4 *>(prm.rgba)[size_t(f) * prm.P + pixel] = toRgba8(cb);
}


}  // namespace sgz
