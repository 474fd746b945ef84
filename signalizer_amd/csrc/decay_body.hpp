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

__device__ __forceinline__ float dbMap(float slope, float st, const DeviceScalars &sc)
{
    const float deltaX = slope * st * sc.minFracRecip;          // :1343 (left-to-right fp32)
    // std::log(float): evaluated in fp64 and rounded once (matches a correctly rounded logf)
    return deltaX > 0.f ? float(log(double(deltaX))) * sc.deltaYRecip : sc.lowerClip;   // :1345
}

// renderSf + the additive blend of one pair's colour into the column buffer, SpectrumDSP.cpp:119-174
__device__ __forceinline__ void blendColour(float (&cb)[3], float intensity, const float *sca, const DeviceScalars &sc)
{
    if (intensity < 0.f) return;
    float colourv[3] = {sca[(NC - 1) * 3 + 0], sca[(NC - 1) * 3 + 1], sca[(NC - 1) * 3 + 2]};
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
#pragma unroll
    for (int c = 0; c < 3; ++c) cb[c] += (1.f - cb[c]) * colourv[c];   // GL_ONE_MINUS_SRC_COLOR
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
                    mag[i] = i <= t ? prm.mapped[size_t(f0 + i) * perFrame + size_t(ps) * prm.P + pixel] : 0.f;
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
                if (prm.lines)
                    prm.lines[(((size_t(f) * prm.C + pair) * G + k) * prm.P + pixel) * 2 + side] = result;
                if (colour) blendColour(cb, result, sca, prm.sc);
            }
        }
    }
    if (prm.rgba) reinterpret_cast<uchar4 *>(prm.rgba)[size_t(f) * prm.P + pixel] = toRgba8(cb);
}


}  // namespace sgz
