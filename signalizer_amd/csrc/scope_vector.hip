// scope_vector.hip -- Oscilloscope and Vectorscope kernels + their C-ABI entry points.  gfx950 only.
//
//  K9  sgz_scope_lanczos_device        drawWavePlot, Lanczos branch  (Source/Oscilloscope/OscilloscopeRendering.cpp:790-891)
//  K10 sgz_scope_zero_crossing_device  ZeroCrossingProcessor::process (Source/Oscilloscope/StreamPreprocessing.h:315-349)
//  K11 sgz_peak_filter_device          runPeakFilter (OscilloscopeDSP.inl:713-886, VectorscopeRendering.cpp:826-889)
//  K12 sgz_vector_polar_device         drawPolarPlot (Source/Vectorscope/VectorscopeRendering.cpp:500-746)
//  K13 sgz_vector_audio_processing_device  Processor::audioProcessing (Source/Vectorscope/Vectorscope.cpp:268-377)
// All are HBM-bound elementwise / scan work: coalesced loads, wave shuffles, no MFMA.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "runtime.hpp"
#include "fade_chain.hpp"

#pragma clang fp contract(off)

using namespace sgz;

namespace {

// ------------------------------------------------------------------------------------------- K9
struct ScopeScalars { double samplePos0, inc, samplesPerPixel, unit0, right, pixelsPerSample; long cursor0; size_t points; };

// triggerMode: OscilloscopeContent::TriggeringMode (0 None, 4 ZeroCrossing)
ScopeScalars scopeDerive(const sgz_scope_view &v, size_t len, uint32_t triggerMode = SGZ_TRIG_ZERO_CROSSING, double cycleSamples = 0.0,
                         double sampleOffset = 0.0, long long transport = 0)
{
    ScopeScalars s{};
    const double horizontalDelta = v.right - v.left;
    const double sizeMinusOne = std::max(1.0, v.window_size - 1);                            // :568
    const double pixelsPerSample = v.rendering_scale * std::fabs((double(v.width) - 1) / (sizeMinusOne * horizontalDelta));   // :572
    s.pixelsPerSample = pixelsPerSample;
    double samplePos;
    if (triggerMode == SGZ_TRIG_WINDOW)
        samplePos = std::ceil(std::fmod(double(transport), v.window_size) - 1);             // :798-801, :814-819
    else if (triggerMode == SGZ_TRIG_ZERO_CROSSING || triggerMode == SGZ_TRIG_ENVELOPE_HOLD)
        samplePos = (v.window_size * 0.5 - double(int(v.window_size * 0.5))) - 1.5;          // triggerState.sampleOffset, OscilloscopeDSP.inl:238
    else if (triggerMode == SGZ_TRIG_SPECTRAL)
        samplePos = cycleSamples * 2 + v.window_size - sampleOffset;                         // :810 (no ceil: :814-819 is None / Window only)
    else
        samplePos = std::ceil(0.0 * 2 + v.window_size - 0.0);                                // :806-816 (cycleSamples = sampleOffset = 0)
    s.inc = horizontalDelta / (v.rendering_scale * (double(v.width) - 1));                   // :822
    s.samplesPerPixel = 1.0 / pixelsPerSample;                                               // :824
    s.unit0 = v.left;
    s.right = v.right;
    s.samplePos0 = samplePos + (-s.unit0 / s.inc * s.samplesPerPixel);                       // :826
    // number of points = trip count of the reference's do-while on a running fp64 sum (:883-889).  It depends on the view
    // only and costs ~1 ns per point on the host, so the last view's count is remembered.
    static thread_local double memo[3] = {0, 0, 0};
    static thread_local size_t memoPoints = 0;
    if (memoPoints && memo[0] == s.unit0 && memo[1] == s.inc && memo[2] == s.right) s.points = memoPoints;
    else {
        size_t n = 0;
        double unitSpacePos = s.unit0;
        do { unitSpacePos += s.inc; ++n; } while (unitSpacePos < (s.right + s.inc));
        s.points = n;
        memo[0] = s.unit0; memo[1] = s.inc; memo[2] = s.right; memoPoints = n;
    }
    if (len) {
        long c = (-long(std::floor(s.samplePos0)) - 10) % long(len);                         // :829, KernelSize = 10
        if (c < 0) c += long(len);
        s.cursor0 = c;
    }
    return s;
}

// cos / sin(pi * i / 10) as libm rounds them (the values std::cos / std::sin return for the double nearest pi*i/10), i = 0 .. 10:
// compile-time constants of the unrolled taps (no table loads in the kernels)
constexpr double kCosPiI10[11] = {1.0, 0.9510565162951535, 0.8090169943749475, 0.5877852522924731, 0.30901699437494745, 6.123233995736766e-17, -0.30901699437494734, -0.587785252292473, -0.8090169943749473, -0.9510565162951535, -1.0};
constexpr double kSinPiI10[11] = {0.0, 0.3090169943749474, 0.5877852522924731, 0.8090169943749475, 0.9510565162951535, 1.0, 0.9510565162951536, 0.8090169943749475, 0.5877852522924732, 0.3090169943749475, 1.2246467991473532e-16};
constexpr double kPi = 3.14159265358979323846;

// sin y for |y| <= pi/2 (Taylor to y^23: truncation 1e-18) and sin / cos y for |y| <= pi/20 (to y^11 / y^12: 4e-17 / 5e-19), Horner in
// y^2: a dozen fused multiply-adds each instead of libm's range reduction.
__device__ __forceinline__ double sinHalfTurn(double y)
{
    const double y2 = y * y;
    double r = 3.8681701706306835e-23;
    r = fma(r, y2, -1.9572941063391263e-20);
    r = fma(r, y2, 8.2206352466243295e-18);
    r = fma(r, y2, -2.8114572543455206e-15);
    r = fma(r, y2, 7.6471637318198164e-13);
    r = fma(r, y2, -1.6059043836821613e-10);
    r = fma(r, y2, 2.5052108385441720e-8);
    r = fma(r, y2, -2.7557319223985893e-6);
    r = fma(r, y2, 1.9841269841269841e-4);
    r = fma(r, y2, -8.3333333333333332e-3);
    r = fma(r, y2, 0.16666666666666666);
    return fma(-(y * y2), r, y);
}
__device__ __forceinline__ void sincosTwentieth(double y, double &sn, double &cs)
{
    const double y2 = y * y;
    double r = -2.5052108385441720e-8;
    r = fma(r, y2, 2.7557319223985893e-6);
    r = fma(r, y2, -1.9841269841269841e-4);
    r = fma(r, y2, 8.3333333333333332e-3);
    r = fma(r, y2, -0.16666666666666666);
    sn = fma(y * y2, r, y);
    double c = 2.0876756987868099e-9;
    c = fma(c, y2, -2.7557319223985888e-7);
    c = fma(c, y2, 2.4801587301587302e-5);
    c = fma(c, y2, -1.3888888888888889e-3);
    c = fma(c, y2, 4.1666666666666664e-2);
    c = fma(c, y2, -0.5);
    cs = fma(c, y2, 1.0);
}

// One output point of drawWavePlot's Lanczos branch: closed form of the reference's running sums (currentSample += spp;
// samplePos += 1 while delta > 1, OscilloscopeRendering.cpp:846-889).  x = 10 + delta in (9, 11]; tap T reads kernel sample
// i = floor(x) - 9 + T, and d_i = x - i = m + e with m = round(x) - i (an integer) and e = x - round(x) in [-1/2, 1/2] (exact):
//   sin(pi d)    = (-1)^m sin(pi e)
//   sin(pi d/10) = sin(pi m/10) cos(pi e/10) + cos(pi m/10) sin(pi e/10)
// Only the m = 0 tap has |d| < 1/2 and it is evaluated directly, so no tap suffers cancellation (a plain angle addition around
// floor(x) loses ~1 % on the nearest tap when x is within 1e-13 of an integer).  round(x) - floor(x) is 0 or 1, so m is one of the two
// compile-time values 9 - T, 10 - T: the table entries are SELECTED between two literals, nothing is loaded and nothing branches.
struct LanczosPoint {
    double x, delta, sPi, s10, c10;
    long fl, shifts;
    bool up;                                                // round(x) == floor(x) + 1
};
__device__ __forceinline__ LanczosPoint lanczosPoint(double samplePos0, double spp, size_t p)
{
    LanczosPoint q;
    const double D = (floor(samplePos0) + double(p) * spp) - samplePos0;
    const double shifts = D > 1.0 ? ceil(D - 1.0) : 0.0;
    q.shifts = long(shifts);
    q.delta = D - shifts;
    q.x = 10.0 + q.delta;
    const double flx = floor(q.x), rnx = rint(q.x);
    q.fl = long(flx);
    q.up = rnx != flx;
    const double e = q.x - rnx;
    q.sPi = sinHalfTurn(kPi * e);
    sincosTwentieth(kPi * e / 10.0, q.s10, q.c10);
    return q;
}
constexpr double signedSinPiI10(int m) { return m < 0 ? -kSinPiI10[-m] : kSinPiI10[m]; }
constexpr double cosPiI10(int m) { return kCosPiI10[m < 0 ? -m : m]; }
// weight of tap T; the reference's loop skips a tap outside the 21-sample kernel window (lanczosTapInside)
__device__ __forceinline__ bool lanczosTapInside(const LanczosPoint &q, int t)
{
    const long i = q.fl - 9 + t;
    return i >= 0 && i < 21;
}
template <int T>
__device__ __forceinline__ double lanczosTap(const LanczosPoint &q)
{
    constexpr int M0 = 9 - T, M1 = 10 - T;
    constexpr double S0 = signedSinPiI10(M0), S1 = signedSinPiI10(M1), C0 = cosPiI10(M0), C1 = cosPiI10(M1);
    const double sm = q.up ? S1 : S0, cm = q.up ? C1 : C0;
    const bool odd = q.up ? bool(M1 & 1) : bool(M0 & 1);
    const bool centre = q.up ? M1 == 0 : M0 == 0;
    const double sa = odd ? -q.sPi : q.sPi;
    const double sb = centre ? q.s10 : fma(sm, q.c10, cm * q.s10);
    const long i = q.fl - 9 + T;
    const double d = q.x - double(i);
    const double pd = kPi * d, pd2 = pd * pd;
    double r = __builtin_amdgcn_rcp(pd2);
    r = fma(fma(-pd2, r, 1.0), r, r);
    r = fma(fma(-pd2, r, 1.0), r, r);
    double wt = (10.0 * sa) * sb * r;
    return d == 0.0 ? 1.0 : wt;
}
template <int T = 0>
__device__ __forceinline__ void lanczosWeights(const LanczosPoint &q, double (&w)[20])
{
    if constexpr (T < 20) {
        w[T] = lanczosTap<T>(q);
        lanczosWeights<T + 1>(q, w);
    }
}

// one thread per output point.  y = sum_i ring[cursor + i] * L(10 + delta - i), L = Lanczos a = 10, fp64.
__global__ void __launch_bounds__(256)
scopeLanczosKernel(const float *ring, size_t len, size_t stride, uint32_t channels, size_t points,
                   double samplePos0, double spp, double unit0, double inc, long cursor0, float2 *xy)
{
    const size_t p = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= points) return;
    const LanczosPoint q = lanczosPoint(samplePos0, spp, p);
    const float ux = float(unit0 + double(p) * inc);
    // the 20 tap weights depend on the point only: computed once, used by every channel
    double w[20];
    lanczosWeights(q, w);
    const long cur = (cursor0 + q.shifts) % long(len);
    long idx = cur + (q.fl - 9);
    if (idx >= long(len)) idx -= long(len);
    uint32_t at[20];
#pragma unroll
    for (int t = 0; t < 20; ++t) {
        at[t] = uint32_t(idx);
        idx = idx + 1 == long(len) ? 0 : idx + 1;
    }
    for (uint32_t c = 0; c < channels; ++c) {
        const float *r = ring + size_t(c) * stride;
        float v[20];
#pragma unroll
        for (int t = 0; t < 20; ++t) v[t] = r[at[t]];
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < 20; ++t) acc = lanczosTapInside(q, t) ? acc + double(v[t]) * w[t] : acc;
        xy[size_t(c) * points + p] = make_float2(ux, float(acc));
    }
}

// ---- drawWavePlot on the handle's front ring (sgz_scope_vertices): the ring's write cursor is read from device memory, the
// sample comes from an evaluator (SampleColourEvaluators.h: one channel, or 0.5 (l +- r)), and a vertex is (x, y, 0) + RGBA8.
__device__ __forceinline__ float evalSample(const float *a, const float *b, uint32_t mode, uint32_t idx)
{
    if (mode == 1u) return 0.5f * (a[idx] + b[idx]);       // MidSideEvaluatorBase<0, std::plus<>>::evaluateSample
    if (mode == 2u) return 0.5f * (a[idx] - b[idx]);       // <1, std::minus<>>
    return a[idx];
}

// The reference's ring of `len` samples inside a physical ring of `cap` >= len (Spectral mode keeps the largest ring the reference can
// ask for, see sgz.h): logical position q (counted from the write cursor = the oldest of the newest `len` samples) -> memory index.
// cap == len (every other mode): (cursor + q) mod len, the ring itself.
__device__ __forceinline__ uint32_t ringPhys(long rel, uint32_t cursor, uint32_t cap, uint32_t len)
{
    long q = rel % long(len);
    if (q < 0) q += long(len);
    return uint32_t((long(cursor) + long(cap - len) + q) % long(cap));
}
// UPixel::lerp(other, t) with a double t (currentColour.lerp(nextColour, delta), OscilloscopeRendering.cpp:876): per component
// (uint8)(a + (b - a) t)
__device__ __forceinline__ uint32_t lerpRgba(uint32_t a, uint32_t b, double t)
{
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double x = double((a >> (8 * k)) & 255u), y = double((b >> (8 * k)) & 255u);
        const double v = x + (y - x) * t;
        const uint32_t c = (!(v > -1.0) || !(v < 256.0)) ? 0u : uint32_t(int(v)) & 255u;
        out |= c << (8 * k);
    }
    return out;
}

// Up to kWaveItems line strips of ONE view in one launch (the two channels of a rendered frame: sgz_scope_vertices_all): the twenty
// fp64 tap weights of a point -- a sine, a sine / cosine pair and twenty divisions -- depend on the view alone and are computed once
// for all strips.  Item fields are separate kernel arguments indexed at compile time (a run-time subscript into a by-value argument
// struct would move it to scratch).
constexpr int kWaveItems = 2;
struct WaveItem { const float *ringA, *ringB; const uint32_t *colRing; float3 *xyz; uint32_t *rgba; uint32_t evalMode, key; };

template <int K>
__global__ void __launch_bounds__(256)
scopeWaveLanczosKernel(const WaveItem it0, const WaveItem it1, uint32_t len, uint32_t cap, const uint32_t *d_cursor,
                       size_t points, double samplePos0, double spp, double unit0, double inc, long cursor0)
{
    const size_t p = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= points) return;
    const uint32_t base = *d_cursor;                        // cursorPosition(): the evaluator's offsets count from it
    const LanczosPoint q = lanczosPoint(samplePos0, spp, p);
    const long idx0 = cursor0 + q.shifts + (q.fl - 9);      // logical position of tap 0
    // ringPhys(idx0 + t, ..) for the twenty taps and the two colour samples: ONE 64-bit modulo per thread (it costs ~100 instructions,
    // and twenty-two of them were most of this kernel once), then steps of one with a wrap -- the same indices
    long q0l = idx0 % long(len);
    if (q0l < 0) q0l += long(len);
    uint32_t lead = base + (cap - len);
    while (lead >= cap) lead -= cap;
    uint32_t at[21];
    {
        uint32_t lq = uint32_t(q0l);
#pragma unroll
        for (int t = 0; t < 21; ++t) {
            const uint32_t ph = lead + lq;
            at[t] = ph >= cap ? ph - cap : ph;
            lq = lq + 1u == len ? 0u : lq + 1u;
        }
    }
    // every sample request of the point in flight before the first is used; the evaluator (SampleColourEvaluators.h) is applied to
    // loaded pairs (mode 0 reads ring A twice: no divergent pointer)
    const float *b0 = it0.evalMode ? it0.ringB : it0.ringA, *b1 = it1.evalMode ? it1.ringB : it1.ringA;
    float va[K][20], vb[K][20];
    auto sampleAt = [](const float *ring, uint32_t index) {     // a 32-bit byte offset from a scalar base (rings are < 2^30 samples)
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ring) + size_t(index * 4u));
    };
#pragma unroll
    for (int t = 0; t < 20; ++t) {
        va[0][t] = sampleAt(it0.ringA, at[t]);
        vb[0][t] = sampleAt(b0, at[t]);
        if constexpr (K > 1) {
            va[1][t] = sampleAt(it1.ringA, at[t]);
            vb[1][t] = sampleAt(b1, at[t]);
        }
    }
    double w[20];
    lanczosWeights(q, w);
    // MidSideEvaluatorBase<0, std::plus<>> / <1, std::minus<>>: 0.5 (a + b), 0.5 (a - b) = 0.5 (a + (-b)); the mode is the same for
    // every thread, and spelled as bit operations so that it costs no branch per tap
    auto evaluated = [](uint32_t mode, float a, float b) {
        const uint32_t flip = mode == 2u ? 0x80000000u : 0u, keep = mode == 0u ? 0xffffffffu : 0u;
        const float half = 0.5f * (a + __uint_as_float(__float_as_uint(b) ^ flip));
        return __uint_as_float((__float_as_uint(a) & keep) | (__float_as_uint(half) & ~keep));
    };
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
#pragma unroll
    for (int t = 0; t < 20; ++t) {
        const bool in = lanczosTapInside(q, t);
        const double s0 = double(evaluated(it0.evalMode, va[0][t], vb[0][t])) * w[t];
        acc[0] = in ? acc[0] + s0 : acc[0];
        if constexpr (K > 1) {
            const double s1 = double(evaluated(it1.evalMode, va[1][t], vb[1][t])) * w[t];
            acc[1] = in ? acc[1] + s1 : acc[1];
        }
    }
    const float ux = float(unit0 + double(p) * inc);
    // colourChannelsByFrequency: the colours of the two newest kernel samples, blended by delta (:836-843, :874-877): logical taps
    // 28 - fl and 29 - fl (cur + 19 = idx0 + (19 - (fl - 9))), fl in {9, 10, 11}
    const uint32_t c0 = q.fl == 11 ? at[17] : q.fl == 10 ? at[18] : at[19];
    const uint32_t c1 = q.fl == 11 ? at[18] : q.fl == 10 ? at[19] : at[20];
    it0.xyz[p] = make_float3(ux, float(acc[0]), 0.f);
    if (it0.rgba) it0.rgba[p] = it0.colRing ? lerpRgba(it0.colRing[c0], it0.colRing[c1], q.delta) : it0.key;
    if constexpr (K > 1) {
        it1.xyz[p] = make_float3(ux, float(acc[1]), 0.f);
        if (it1.rgba) it1.rgba[p] = it1.colRing ? lerpRgba(it1.colRing[c0], it1.colRing[c1], q.delta) : it1.key;
    }
}

// Linear: vertex i = (i, sample[cursor - bufferOffset + i], 0), i < endCondition (OscilloscopeRendering.cpp:707-741)
__global__ void __launch_bounds__(256)
scopeWaveLinearKernel(const float *ringA, const float *ringB, uint32_t evalMode, uint32_t len, uint32_t cap, const uint32_t *d_cursor,
                      size_t points, long start0, uint32_t key, const uint32_t *colRing, float3 *xyz, uint32_t *rgba)
{
    const size_t p = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= points) return;
    const uint32_t idx = ringPhys(start0 + long(p), *d_cursor, cap, len);
    xyz[p] = make_float3(float(p), evalSample(ringA, ringB, evalMode, idx), 0.f);
    if (rgba) rgba[p] = colRing ? colRing[idx] : key;      // :664-677: drawer.addColour(data.second) per sample
}

// Rectangular (:746-789): sample i becomes the two vertices (i, y), (i + 1, y), coloured with the previous sample's colour (the first:
// its own, evaluator.evaluateColour() before the loop) and its own
__global__ void __launch_bounds__(256)
scopeWaveRectKernel(const float *ringA, const float *ringB, uint32_t evalMode, uint32_t len, uint32_t cap, const uint32_t *d_cursor,
                    size_t samples, long start0, uint32_t key, const uint32_t *colRing, float3 *xyz, uint32_t *rgba)
{
    const size_t p = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= samples) return;
    const uint32_t idx = ringPhys(start0 + long(p), *d_cursor, cap, len);
    const float y = evalSample(ringA, ringB, evalMode, idx);
    xyz[2 * p] = make_float3(float(p), y, 0.f);
    xyz[2 * p + 1] = make_float3(float(p) + 1.f, y, 0.f);
    if (rgba) {
        const uint32_t prev = p ? ringPhys(start0 + long(p) - 1, *d_cursor, cap, len) : idx;
        rgba[2 * p] = colRing ? colRing[prev] : key;
        rgba[2 * p + 1] = colRing ? colRing[idx] : key;
    }
}

// ------------------------------------------------------------------------------------------- K10
// One workgroup scans the whole block: each thread owns a contiguous segment.
struct ZcResult { unsigned long long lastArm; unsigned long long count; int armed; int anyArm; double lastSample; };

__device__ __forceinline__ double zcSample(uint32_t mode, const float *a, const float *b, size_t i)
{
    switch (mode) {
    case SGZ_OSC_MID: case SGZ_OSC_MIDSIDE: return double(0.5f * (a[i] + b[i]));   // OscilloscopeDSP.inl:371-376
    case SGZ_OSC_SIDE: return double(0.5f * (a[i] - b[i]));                         // :377-382
    default: return double(a[i]);
    }
}

// Sequential automaton (StreamPreprocessing.h:331-347):  arm_i = (s_i > 0 && s_{i-1} < 0) sets armed and
// crossOrigin = clock + i;  then if (armed && s_i > threshold) { armed = false; push(crossOrigin); }.
// Closed form: with lastArm(i) = max{j <= i : arm_j} and lastThr(i) = max{k <= i : s_k > threshold},
//   fire_i  <=>  s_i > threshold  &&  lastArm(i) > lastThr(i-1),   value = clock + lastArm(i)
// (virtual indices: an arm inherited from the previous block sits at -1, "none" at -3, no threshold yet -2).
// So the trigger list is two prefix-max scans + an ordered compaction.
__global__ void __launch_bounds__(1024)
zeroCrossingKernel(uint32_t mode, const float *a, const float *b, size_t n, double prevState, double threshold,
                   int armedIn, unsigned long long originIn, unsigned long long clockPlusCount,
                   unsigned long long *out, size_t maxOut, ZcResult *res)
{
    __shared__ long long sArm[1024], sThr[1024];
    __shared__ unsigned int sCnt[1024];
    const int tid = threadIdx.x, T = blockDim.x;
    const size_t seg = (n + T - 1) / T;
    const size_t i0 = min(n, size_t(tid) * seg), i1 = min(n, i0 + seg);
    long long segArm = -4, segThr = -4;                 // -4: no event in this segment
    for (size_t i = i0; i < i1; ++i) {
        const double s = zcSample(mode, a, b, i);
        const double prev = i ? zcSample(mode, a, b, i - 1) : prevState;
        if (s > 0 && prev < 0) segArm = (long long)i;
        if (s > threshold) segThr = (long long)i;
    }
    sArm[tid] = segArm; sThr[tid] = segThr;
    __syncthreads();
    if (tid == 0) {                                     // exclusive prefix-max over the 1024 segments
        long long ca = armedIn ? -1 : -3, ct = -2;
        for (int t = 0; t < T; ++t) {
            const long long la = sArm[t], lt = sThr[t];
            sArm[t] = ca; sThr[t] = ct;
            if (la > -4) ca = la;
            if (lt > -4) ct = lt;
        }
    }
    __syncthreads();
    const long long inArm = sArm[tid], inThr = sThr[tid];
    long long la = inArm, lt = inThr;
    unsigned int fires = 0;
    for (size_t i = i0; i < i1; ++i) {                  // count
        const double s = zcSample(mode, a, b, i);
        const double prev = i ? zcSample(mode, a, b, i - 1) : prevState;
        if (s > 0 && prev < 0) la = (long long)i;
        if (s > threshold) { if (la > lt) ++fires; lt = (long long)i; }
    }
    sCnt[tid] = fires;
    __syncthreads();
    if (tid == 0) {
        unsigned int acc = 0;
        for (int t = 0; t < T; ++t) { const unsigned int c = sCnt[t]; sCnt[t] = acc; acc += c; }
        res->count = acc;
    }
    __syncthreads();
    size_t pos = sCnt[tid];
    la = inArm; lt = inThr;
    for (size_t i = i0; i < i1; ++i) {                  // ordered write
        const double s = zcSample(mode, a, b, i);
        const double prev = i ? zcSample(mode, a, b, i - 1) : prevState;
        if (s > 0 && prev < 0) la = (long long)i;
        if (s > threshold) {
            if (la > lt) {
                if (pos < maxOut) out[pos] = (la == -1) ? originIn : clockPlusCount + (unsigned long long)la;
                ++pos;
            }
            lt = (long long)i;
        }
    }
    if (tid == T - 1) {                                 // state after the whole block
        res->armed = (la > lt) ? 1 : 0;
        res->anyArm = la >= 0 ? 1 : 0;
        res->lastArm = la >= 0 ? (unsigned long long)la : 0ull;
        res->lastSample = n ? zcSample(mode, a, b, n - 1) : prevState;
    }
}

// ------------------------------------------------------------------------------------------- K11
__global__ void __launch_bounds__(1024)
peakKernel(const float *ch, size_t stride, size_t stop, float *peaks)
{
    __shared__ float sm[16];
    const float *x = ch + size_t(blockIdx.x) * stride;
    float m = 0.f;
    for (size_t i = threadIdx.x; i < stop; i += blockDim.x) m = fmaxf(m, fabsf(x[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) r = fmaxf(r, sm[w]);
        peaks[blockIdx.x] = r;
    }
}

// ------------------------------------------------------------------------------------------- K12
// The reference's fade ramp is a running fp32 sum per SIMD lane (vSampleFade += fadePerSample * V once per iteration,
// VectorscopeRendering.cpp:528-543,:592): sequential as written, but it only depends on (n, lanes).  One small kernel replays it
// into ramp[k][lane] = outFade of iteration k, lane; the per-sample kernel then just looks its value up.  The replay is not a walk:
// one thread per lane finds the sum's arithmetic progressions (fade_chain.hpp), all threads evaluate them; only a chain with ties at
// every step is walked addition by addition.
__global__ void __launch_bounds__(256) fadeRampKernel(size_t n, uint32_t lanes, long iters, float *ramp /*[iters][lanes]*/)
{
    constexpr uint32_t kLanes = 16;
    __shared__ FadeSeg segs[kLanes][kFadeSegs];
    __shared__ int nseg[kLanes], overflow;
    const uint32_t tid = threadIdx.x;
    const float fadePerSample = 1.0f / float(n);
    const float incr = fadePerSample * float(lanes);
    if (tid == 0) overflow = (lanes > kLanes || iters > 0x7fffffffL) ? 1 : 0;
    __syncthreads();
    if (tid < lanes && !overflow) {
        float f = fadePerSample * float(tid), last = 0.f;
        int cnt = 0;
        if (!fadeChainSegments(f, incr, uint32_t(iters), segs[tid], cnt, last)) atomicExch(&overflow, 1);
        nseg[tid] = cnt;
    }
    __syncthreads();
    if (!overflow) {
        fadeChainFill(&segs[0][0], nseg, lanes, uint32_t(iters), tid, 256u, -1.0f, ramp);
        return;
    }
    if (tid < lanes) {                                        // (the reference's walk: `iters` dependent additions per lane)
        float f = fadePerSample * float(tid);
        for (long k = 0; k < iters; ++k) {
            ramp[k * long(lanes) + tid] = f - 1.0f;
            f += incr;
        }
    }
}

__global__ void __launch_bounds__(256)
vectorPolarKernel(const float *planar, size_t stride, uint32_t pairs, size_t n, uint32_t lanes, long iters, const float *ramp,
                  float3 *xyz)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t pair = blockIdx.y;
    if (i >= n) return;
    const float l = planar[size_t(2 * pair) * stride + i];
    const float r = planar[size_t(2 * pair + 1) * stride + i];
    const float cosineRotation = -0.70710678118654752440f, sineRotation = 0.70710678118654752440f;   // :521-524
    const float length = fmaxf(fabsf(l), fabsf(r));                                                 // :563
    const float vY = l * cosineRotation - r * sineRotation;                                          // :566
    const float vX = l * sineRotation + r * cosineRotation;                                          // :567
    float angle = atanf(vX / vY);                                                                    // :576
    if (l == 0.f && r == 0.f) angle = 0.f;                                                           // :578
    float sx, cy;
    sincosf(angle, &sx, &cy);
    // fade ramp: SIMD body for i < mainEnd (table above), scalar tail afterwards (:600-634)
    const float fadePerSample = 1.0f / float(n);
    const long V = long(lanes);
    const long mainEnd = iters * V;
    float fade;
    if (long(i) < mainEnd) {
        fade = ramp[i];                                       // ramp[k * V + lane], k = i / V, lane = i % V
    } else {
        // outFade[vectorLength-1] of the last SIMD iteration; never written when there was none: initial ramp value (:535-538)
        const float base = iters > 0 ? ramp[size_t(iters - 1) * V + (V - 1)] : fadePerSample * float(V - 1);
        fade = base - float(long(i) - mainEnd) * fadePerSample;
    }
    xyz[size_t(pair) * n + i] = make_float3(sx * length, cy * length, fade);
}

// ------------------------------------------------------------------------------------------- K13
struct VecState { float env[2]; float bal[2][2]; float phase[2]; };

__global__ void __launch_bounds__(64)
vectorAudioKernel(const float *L, const float *R, size_t n, float envelope, float pole0, float pole1, VecState *st)
{
    __shared__ float sL[64], sR[64], sP[64];
    const int lane = threadIdx.x;
    // lane k < 8 owns one recurrence: 0,1 env L/R; 2,3 slow balance L/R; 4,5 fast balance L/R; 6 slow phase; 7 fast phase
    float y = 0.f, a = 0.f;
    if (lane < 8) {
        const float *s = reinterpret_cast<const float *>(st);
        y = s[lane];
        a = lane < 2 ? envelope : ((lane == 2 || lane == 3 || lane == 6) ? pole0 : pole1);
    }
    const int sel = (lane == 0 || lane == 2 || lane == 4) ? 0 : ((lane == 1 || lane == 3 || lane == 5) ? 1 : 2);
    for (size_t base = 0; base < n; base += 64) {
        const size_t i = base + lane;
        if (i < n) {
            const float l = L[i], r = R[i];
            const float mReal = -0.70710678118654752440f, mImag = 0.70710678118654752440f;
            const float vX = l * mReal - r * mImag;                        // Vectorscope.cpp:303
            const float vY = r * mImag + l * mReal;                        // :304
            const float radians = atanf(vY / vX);                          // :310
            const float ang = (vX == 0.f && vY == 0.f) ? 0.78539816339744830962f : radians;   // :311
            sP[lane] = cosf(ang * 2.0f);                                   // :316
            sL[lane] = l * l; sR[lane] = r * r;                            // :323-324
        }
        __syncthreads();
        if (lane < 8) {
            const float *src = sel == 0 ? sL : (sel == 1 ? sR : sP);
            const int m = int(min(size_t(64), n - base));
            for (int k = 0; k < m; ++k) { const float x = src[k]; y = x + a * (y - x); }   // :327-342
        }
        __syncthreads();
    }
    if (lane < 8) reinterpret_cast<float *>(st)[lane] = y;
}

// Scratch of the stage calls: stream-ordered allocations (hipMallocAsync / hipFreeAsync on the caller's stream), so the library keeps
// no process-wide device state -- any number of host threads, streams and devices may use the stage calls at once.
struct StreamScratch {
    void *p = nullptr;
    hipStream_t s;
    explicit StreamScratch(hipStream_t stream) : s(stream) {}
    hipError_t get(size_t bytes) { return hipMallocAsync(&p, bytes, s); }
    ~StreamScratch() { if (p) (void)hipFreeAsync(p, s); }
    StreamScratch(const StreamScratch &) = delete;
    StreamScratch &operator=(const StreamScratch &) = delete;
};

// which branch drawWavePlot takes: Lanczos falls back to Linear below one pixel per sample (:575-578)
bool waveIsLanczos(const sgz_scope_view &v, uint32_t interpolation)
{
    if (interpolation != SGZ_SUBSAMPLE_LANCZOS) return false;
    const double sizeMinusOne = std::max(1.0, v.window_size - 1);
    const double pps = v.rendering_scale * std::fabs((double(v.width) - 1) / (sizeMinusOne * (v.right - v.left)));
    return !(pps < 1);
}

// Rectangular falls back to Linear below one pixel per sample like Lanczos does; None never does (:575-578)
bool waveIsRect(const sgz_scope_view &v, uint32_t interpolation)
{
    if (interpolation != SGZ_SUBSAMPLE_RECTANGULAR) return false;
    const double sizeMinusOne = std::max(1.0, v.window_size - 1);
    const double pps = v.rendering_scale * std::fabs((double(v.width) - 1) / (sizeMinusOne * (v.right - v.left)));
    return !(pps < 1);
}

}  // namespace

namespace sgz {

size_t scopeVertexCount(const sgz_scope_view &view, uint32_t interpolation, uint32_t triggerMode, double cycleSamples)
{
    if (waveIsLanczos(view, interpolation)) return scopeDerive(view, 0).points;
    // endCondition = roundedWindow + quantizedCycleSamples (:614, :631)
    const long quantizedCycleSamples = triggerMode == SGZ_TRIG_SPECTRAL ? long(std::ceil(cycleSamples)) : 0;
    const size_t samples = size_t(std::max<long>(2, long(std::ceil(view.window_size))) + quantizedCycleSamples);
    return waveIsRect(view, interpolation) ? 2 * samples : samples;
}

// Two Lanczos strips of one view in ONE launch (shared tap weights); false when the view's strips are not Lanczos (the caller then
// launches them one by one)
bool launchScopeLanczosPair(const sgz_scope_view &view, uint32_t triggerMode, uint32_t interpolation, const float *const ringA[2],
                            const float *const ringB[2], const uint32_t evalMode[2], uint32_t size, uint32_t cap, const uint32_t *d_cursor,
                            double cycleSamples, double sampleOffset, long long transport, const uint32_t key[2], const uint32_t *const colRing[2],
                            float *const d_xyz[2], uint32_t *const d_rgba[2], size_t capacity, size_t *points, hipStream_t stream, hipError_t *err)
{
    if (!waveIsLanczos(view, interpolation)) return false;
    const int block = 256;
    const ScopeScalars s = scopeDerive(view, size, triggerMode, cycleSamples, sampleOffset, transport);
    if (s.points > capacity) { *err = hipErrorInvalidValue; return true; }
    const WaveItem a{ringA[0], ringB[0], colRing[0], reinterpret_cast<float3 *>(d_xyz[0]), d_rgba[0], evalMode[0], key[0]};
    const WaveItem b{ringA[1], ringB[1], colRing[1], reinterpret_cast<float3 *>(d_xyz[1]), d_rgba[1], evalMode[1], key[1]};
    hipLaunchKernelGGL(scopeWaveLanczosKernel<2>, dim3(unsigned((s.points + block - 1) / block)), dim3(block), 0, stream, a, b, size, cap,
                       d_cursor, s.points, s.samplePos0, s.samplesPerPixel, s.unit0, s.inc, s.cursor0);
    *points = s.points;
    *err = hipGetLastError();
    return true;
}

// ringA / ringB / colRing: one channel's plane of the physical ring (`cap` elements); `size`: the reference's ring of the moment
hipError_t launchScopeVertices(const sgz_scope_view &view, uint32_t triggerMode, uint32_t interpolation, const float *ringA,
                               const float *ringB, uint32_t evalMode, uint32_t size, uint32_t cap, const uint32_t *d_cursor,
                               double cycleSamples, double sampleOffset, long long transport, uint32_t rgba, const uint32_t *colRing,
                               float *d_xyz, uint32_t *d_rgba, size_t capacity, size_t *points, hipStream_t stream)
{
    const int block = 256;
    if (waveIsLanczos(view, interpolation)) {
        const ScopeScalars s = scopeDerive(view, size, triggerMode, cycleSamples, sampleOffset, transport);
        if (s.points > capacity) return hipErrorInvalidValue;
        const WaveItem it{ringA, ringB, colRing, reinterpret_cast<float3 *>(d_xyz), d_rgba, evalMode, rgba};
        hipLaunchKernelGGL(scopeWaveLanczosKernel<1>, dim3(unsigned((s.points + block - 1) / block)), dim3(block), 0, stream, it, it,
                           size, cap, d_cursor, s.points, s.samplePos0, s.samplesPerPixel, s.unit0, s.inc, s.cursor0);
        *points = s.points;
    } else {
        const long roundedWindow = long(std::ceil(view.window_size));
        long bufferOffset, quantizedCycleSamples = 0;
        if (triggerMode == SGZ_TRIG_WINDOW) {
            bufferOffset = long(std::ceil(std::fmod(double(transport), view.window_size)));   // :588-592
        } else if (triggerMode == SGZ_TRIG_ZERO_CROSSING || triggerMode == SGZ_TRIG_ENVELOPE_HOLD) {
            const double realOffset = (view.window_size * 0.5 - double(int(view.window_size * 0.5))) - 1.5;
            bufferOffset = long(std::ceil(realOffset));                                 // :593-594
        } else {
            // :598-612; this branch is never Lanczos, so cycleBuffers = 1
            if (triggerMode == SGZ_TRIG_SPECTRAL) quantizedCycleSamples = long(std::ceil(cycleSamples));
            bufferOffset = roundedWindow + quantizedCycleSamples;
        }
        const size_t n = size_t(std::max<long>(2, roundedWindow) + quantizedCycleSamples);
        if (waveIsRect(view, interpolation)) {
            if (2 * n > capacity) return hipErrorInvalidValue;
            hipLaunchKernelGGL(scopeWaveRectKernel, dim3(unsigned((n + block - 1) / block)), dim3(block), 0, stream, ringA, ringB, evalMode,
                               size, cap, d_cursor, n, -bufferOffset, rgba, colRing, reinterpret_cast<float3 *>(d_xyz), d_rgba);
            *points = 2 * n;
            return hipGetLastError();
        }
        if (n > capacity) return hipErrorInvalidValue;
        hipLaunchKernelGGL(scopeWaveLinearKernel, dim3(unsigned((n + block - 1) / block)), dim3(block), 0, stream, ringA, ringB, evalMode,
                           size, cap, d_cursor, n, -bufferOffset, rgba, colRing, reinterpret_cast<float3 *>(d_xyz), d_rgba);
        *points = n;
    }
    return hipGetLastError();
}

}  // namespace sgz

extern "C" {

size_t sgz_scope_num_points(const sgz_scope_view *view)
{
    if (!view || view->width < 2 || !(view->right > view->left)) return 0;
    return scopeDerive(*view, 0).points;
}

sgz_status sgz_scope_lanczos_device(const sgz_scope_view *view, const float *d_ring, size_t len, size_t stride,
                                    uint32_t channels, float *d_xy, void *stream)
{
    if (!view || !d_ring || !d_xy || len == 0 || view->width < 2 || !(view->right > view->left))
        return fail(SGZ_EINVAL, "bad scope arguments");
    const ScopeScalars s = scopeDerive(*view, len);
    const int block = 256;
    const unsigned grid = unsigned((s.points + block - 1) / block);
    hipLaunchKernelGGL(scopeLanczosKernel, dim3(grid), dim3(block), 0, reinterpret_cast<hipStream_t>(stream), d_ring, len,
                       stride, channels, s.points, s.samplePos0, s.samplesPerPixel, s.unit0, s.inc, s.cursor0,
                       reinterpret_cast<float2 *>(d_xy));
    SGZ_HIP(hipGetLastError());
    return SGZ_OK;
}

sgz_status sgz_scope_zero_crossing_device(sgz_zero_crossing_state *zs, uint32_t osc_mode, const float *d_a, const float *d_b,
                                          size_t n, uint64_t *d_triggers, size_t max_triggers, size_t *num_triggers,
                                          void *stream)
{
    if (!zs || !d_a || !d_triggers || !num_triggers) return fail(SGZ_EINVAL, "null argument");
    if (!d_b) d_b = d_a;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n == 0) { *num_triggers = 0; return SGZ_OK; }
    StreamScratch scr(s);
    SGZ_HIP(scr.get(sizeof(ZcResult)));
    ZcResult *d_res = reinterpret_cast<ZcResult *>(scr.p);
    hipLaunchKernelGGL(zeroCrossingKernel, dim3(1), dim3(1024), 0, s, osc_mode, d_a, d_b, n, zs->state, zs->threshold,
                       int(zs->armed), (unsigned long long)zs->cross_origin,
                       (unsigned long long)(zs->steady_clock + zs->count), reinterpret_cast<unsigned long long *>(d_triggers),
                       max_triggers, d_res);
    SGZ_HIP(hipGetLastError());
    ZcResult h{};
    SGZ_HIP(hipMemcpyAsync(&h, d_res, sizeof(h), hipMemcpyDeviceToHost, s));
    SGZ_HIP(hipStreamSynchronize(s));
    *num_triggers = size_t(h.count);
    if (h.anyArm) zs->cross_origin = zs->steady_clock + zs->count + h.lastArm;
    zs->armed = h.armed;
    zs->state = h.lastSample;
    zs->count += n;
    return SGZ_OK;
}

sgz_status sgz_peak_filter_device(const float *d_ch, size_t stride, uint32_t channels, size_t n, uint32_t lanes,
                                  double coeff_pow, double *env, double *gain, void *stream)
{
    if (!d_ch || !env || !gain || channels == 0 || lanes == 0 || (lanes & (lanes - 1))) return fail(SGZ_EINVAL, "bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    StreamScratch scr(s);
    SGZ_HIP(scr.get(sizeof(float) * channels + 64));
    float *d_peaks = reinterpret_cast<float *>(scr.p);
    const size_t stop = n - (n & size_t(lanes - 1));          // SIMD tail dropped, :740 / :860
    hipLaunchKernelGGL(peakKernel, dim3(channels), dim3(1024), 0, s, d_ch, stride, stop, d_peaks);
    SGZ_HIP(hipGetLastError());
    std::vector<float> peaks(channels);
    SGZ_HIP(hipMemcpyAsync(peaks.data(), d_peaks, sizeof(float) * channels, hipMemcpyDeviceToHost, s));
    SGZ_HIP(hipStreamSynchronize(s));
    double start = 0;
    for (uint32_t c = 0; c < channels; ++c) {                  // VectorscopeRendering.cpp:873-879
        const double highest = double(peaks[c]);
        const float e = float(std::max(double(float(env[c])) * coeff_pow, highest * highest));
        env[c] = double(e);
        start = std::max(start, std::sqrt(double(e)));
    }
    *gain = 1.0 / start;
    return SGZ_OK;
}

sgz_status sgz_vector_polar_device(const float *d_planar, size_t stride, uint32_t pairs, size_t n, uint32_t lanes,
                                   float *d_xyz, void *stream)
{
    if (!d_planar || !d_xyz || pairs == 0 || lanes == 0) return fail(SGZ_EINVAL, "bad argument");
    if (n == 0) return SGZ_OK;
    if (lanes > 64) return fail(SGZ_EINVAL, "lanes > 64");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long iters = (long(n) > long(lanes)) ? (long(n) - 1) / long(lanes) : 0;      // VectorscopeRendering.cpp:528
    // the ramp table depends on (n, lanes) only: the first call with a pair of them builds it (and waits for it once, so that any
    // stream may read it afterwards), later calls find it -- a few tables per device are kept for the life of the process; beyond
    // that the call rebuilds its own in stream order as it always did
    StreamScratch ramp(s);
    float *d_ramp = nullptr;
    {
        struct Entry { int device; size_t n; uint32_t lanes; float *table; };
        static std::mutex mu;
        static std::vector<Entry> cache;
        int device = 0;
        SGZ_HIP(hipGetDevice(&device));
        std::lock_guard<std::mutex> lk(mu);
        for (const Entry &e : cache)
            if (e.device == device && e.n == n && e.lanes == lanes) { d_ramp = e.table; break; }
        if (!d_ramp && cache.size() < 8) {
            float *t = nullptr;
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&t), (size_t(iters) * lanes + 1) * sizeof(float)));
            if (iters > 0) hipLaunchKernelGGL(fadeRampKernel, dim3(1), dim3(256), 0, s, n, lanes, iters, t);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) { (void)hipFree(t); return hipFail(e, "fadeRampKernel launch"); }   // (never cache a table that was not built)
            if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess) { (void)hipFree(t); return hipFail(e, "hipStreamSynchronize"); }
            cache.push_back(Entry{device, n, lanes, t});
            d_ramp = t;
        }
    }
    if (!d_ramp) {
        SGZ_HIP(ramp.get((size_t(iters) * lanes + 1) * sizeof(float)));
        d_ramp = reinterpret_cast<float *>(ramp.p);
        if (iters > 0) hipLaunchKernelGGL(fadeRampKernel, dim3(1), dim3(256), 0, s, n, lanes, iters, d_ramp);
    }
    const int block = 256;
    dim3 grid(unsigned((n + block - 1) / block), pairs);
    hipLaunchKernelGGL(vectorPolarKernel, grid, dim3(block), 0, s, d_planar, stride, pairs, n, lanes, iters, d_ramp,
                       reinterpret_cast<float3 *>(d_xyz));
    SGZ_HIP(hipGetLastError());
    return SGZ_OK;
}

sgz_status sgz_vector_audio_processing_device(sgz_vector_filters *f, const float *d_left, const float *d_right, size_t n,
                                              uint32_t lanes, float envelope_coeff, float stereo_coeff, float second_speed,
                                              int env_mode, float *gain_out, void *stream)
{
    if (!f || !d_left || !d_right || lanes == 0 || (lanes & (lanes - 1))) return fail(SGZ_EINVAL, "bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    StreamScratch scratch(s);
    SGZ_HIP(scratch.get(sizeof(VecState)));
    void *scr = scratch.p;
    VecState h{};
    h.env[0] = f->env[0]; h.env[1] = f->env[1];
    // lane order of the kernel: env L,R ; slow bal L,R ; fast bal L,R ; slow phase ; fast phase
    h.bal[0][0] = f->balance[0][0]; h.bal[0][1] = f->balance[0][1];
    h.bal[1][0] = f->balance[1][0]; h.bal[1][1] = f->balance[1][1];
    h.phase[0] = f->phase[0]; h.phase[1] = f->phase[1];
    SGZ_HIP(hipMemcpyAsync(scr, &h, sizeof(h), hipMemcpyHostToDevice, s));
    n -= n & size_t(lanes - 1);                                               // Vectorscope.cpp:292
    const float pole1 = std::pow(stereo_coeff, second_speed);                 // :281
    if (n) {
        hipLaunchKernelGGL(vectorAudioKernel, dim3(1), dim3(64), 0, s, d_left, d_right, n, envelope_coeff, stereo_coeff, pole1,
                           reinterpret_cast<VecState *>(scr));
        SGZ_HIP(hipGetLastError());
    }
    SGZ_HIP(hipMemcpyAsync(&h, scr, sizeof(h), hipMemcpyDeviceToHost, s));
    SGZ_HIP(hipStreamSynchronize(s));
    if (env_mode == 1) {                                                      // :346-363
        const double currentEnvelope = 1.0 / double(std::max(std::sqrt(h.env[0]), std::sqrt(h.env[1])));   // std::sqrt(T), T = float (:351)
        f->env[0] = h.env[0]; f->env[1] = h.env[1];
        if (std::isnormal(currentEnvelope) && gain_out) *gain_out = float(currentEnvelope);
    }
    f->balance[0][0] = h.bal[0][0]; f->balance[0][1] = h.bal[0][1];
    f->balance[1][0] = h.bal[1][0]; f->balance[1][1] = h.bal[1][1];
    f->phase[0] = h.phase[0]; f->phase[1] = h.phase[1];
    return SGZ_OK;
}

}  // extern "C"
