// tracker.hip -- the frequency tracker's peak search over the raw transform: the raw-FFT branch of
// Spectrum::drawFrequencyTracking, Source/Spectrum/SpectrumRendering.cpp:379-469.  gfx950 only.
//   nearest peak of |source|^2 inside the +-3 % neighbourhood of the mouse position (std::max_element: the FIRST largest), the walk
//   along a still rising edge when the peak sits on a boundary of the range (:400-427), then the parabolic fit through the three dB
//   values around it (:431-444) -> bin, fractional bin, frequency, dB.
// One workgroup: the range is at most a few thousand bins; the reduction key is (square, smaller index wins), i.e. max_element's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "runtime.hpp"

#pragma clang fp contract(off)

using namespace sgz;

namespace {

__global__ void __launch_bounds__(1024)
trackPeakKernel(const float *bins, uint32_t N, long lower, long higher, float invSize, double sampleRate, sgz_peak *out)
{
    __shared__ float sSq[16];
    __shared__ long sIdx[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto sqOf = [&](long k) { const float m = bins[k]; return m * m + 0.f; };         // Math::square(complex) with imag == 0
    float best = -1.f;
    long arg = higher + 1;
    for (long k = lower + tid; k <= higher; k += blockDim.x) {
        const float s = sqOf(k);
        if (s > best) { best = s; arg = k; }                    // ascending k per thread: its first largest (a NaN square never wins)
    }
    // max_element's order over the whole range: larger square, then the smaller index
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const long oa = __shfl_xor(arg, o);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) { sSq[wave] = best; sIdx[wave] = arg; }
    __syncthreads();
    if (tid != 0) return;
    for (unsigned w = 1; w < blockDim.x / 64; ++w)
        if (sSq[w] > best || (sSq[w] == best && sIdx[w] < arg)) { best = sSq[w]; arg = sIdx[w]; }
    long peak = arg > higher ? lower : arg;                     // (all squares NaN: max_element keeps the first element)
    if (peak == lower && lower != 0) {                          // :400-413
        for (;;) {
            const long next = peak - 1;
            if (next == 0) break;
            else if (sqOf(next) < sqOf(peak)) break;
            else peak = next;
        }
    } else if (peak == higher - 1) {                            // :414-427
        for (;;) {
            const long next = peak + 1;
            if (next == long(N)) break;
            else if (sqOf(next) < sqOf(peak)) break;
            else peak = next;
        }
    }
    const long ia = peak == 0 ? 0 : peak - 1, ic = peak == long(N) ? peak : peak + 1;
    const float alpha = 20 * log10f(fabsf(bins[ia] * invSize));
    const float beta = 20 * log10f(fabsf(bins[peak] * invSize));
    const float gamma = 20 * log10f(fabsf(bins[ic] * invSize));
    const double phi = 0.5 * (alpha - gamma) / (alpha - 2 * beta + gamma);
    auto isNormal = [](double v) { const double a = fabs(v); return a >= 2.2250738585072014e-308 && a < INFINITY; };
    const double peakFraction = 2 * (double(peak) + (isNormal(phi) ? phi : 0)) / double(N);
    double peakDBs = beta - 0.25 * (alpha - gamma) * phi;
    if (!isNormal(peakDBs)) peakDBs = 20 * log10(double(fabsf(bins[peak])) / (double(N) * 0.5));
    out->peak_offset = double(peak); out->peak_fraction = peakFraction; out->peak_frequency = 0.5 * peakFraction * sampleRate;
    out->peak_dbs = peakDBs; out->alpha = alpha; out->beta = beta; out->gamma = gamma; out->phi = phi;
}

}  // namespace

namespace sgz {

// bounds of the search range from the mouse position (:383-392), then the kernel; d_out: DEVICE sgz_peak
sgz_status runTrackPeak(const Plan &p, const float *d_bins, double mouseFraction, sgz_peak *d_out, hipStream_t stream)
{
    if (p.cfg.channel_mode == SGZ_CH_PHASE || p.cfg.channel_mode == SGZ_CH_COMPLEX)
        return fail(SGZ_EUNSUPPORTED, "frequency tracker: raw-FFT branch of the non-Complex magnitude modes (SpectrumRendering.cpp:301)");
    if (!std::isfinite(mouseFraction)) return fail(SGZ_EINVAL, "mouse_fraction");
    mouseFraction = mouseFraction < 0 ? 0 : (mouseFraction > 1 ? 1 : mouseFraction);                      // :292
    const double nearby = 0.03, sampleRate = double(p.cfg.sample_rate);
    const long points = long(p.P), N = long(p.N);
    auto confine = [](long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); };
    long lower = std::llround(double(points) * (mouseFraction - nearby));
    lower = std::llround(double(float(size_t(N)) * p.mapped[size_t(confine(lower, 0, points - 1))]) / sampleRate);
    long higher = std::llround(double(points) * (mouseFraction + nearby));
    higher = std::llround(double(float(size_t(N)) * p.mapped[size_t(confine(higher, 0, points - 1))]) / sampleRate);
    lower = confine(lower, 0, N); higher = confine(higher, 0, N);
    hipLaunchKernelGGL(trackPeakKernel, dim3(1), dim3(1024), 0, stream, d_bins, p.N, lower, higher, p.scalars.invSize, sampleRate, d_out);
    SGZ_HIP(hipGetLastError());
    return SGZ_OK;
}

}  // namespace sgz

namespace sgz {

// The tracker's line-results branch (SpectrumRendering.cpp:300-377), host arithmetic on host-resident results (see sgz.h).
sgz_status trackPeakLines(const Plan &p, const float *results, double mouseFraction, sgz_line_peak *out)
{
    if (!std::isfinite(mouseFraction)) return fail(SGZ_EINVAL, "mouse_fraction");
    mouseFraction = mouseFraction < 0 ? 0 : (mouseFraction > 1 ? 1 : mouseFraction);                      // :292
    const double nearbyFractionToConsider = 0.03;
    const size_t N = p.P;                                                                                 // results.size()
    if (N == 0) return fail(SGZ_EINVAL, "no axis points");
    auto left = [&](size_t i) { return results[2 * i]; };                                                 // UComplex::leftMagnitude
    const size_t pivot = size_t(std::llround(double(N) * mouseFraction));
    const size_t range = size_t(std::llround(double(N) * nearbyFractionToConsider));
    const size_t lowerBound = range > pivot ? 0 : pivot - range;
    const size_t higherBound = range + pivot > N ? N : range + pivot;
    // std::max_element over [lowerBound, higherBound): the first largest (an empty range -- fewer than 17 axis points -- yields its own
    // begin in the reference; confined to the last point here so that nothing is read behind the results)
    size_t peak = lowerBound < N ? lowerBound : N - 1;
    for (size_t i = lowerBound + 1; i < higherBound; ++i)
        if (left(peak) < left(i)) peak = i;
    if (peak == lowerBound && lowerBound != 0) {                                                          // :320-332
        for (;;) {
            const size_t next = peak - 1;
            if (next == 0) break;
            else if (left(next) < left(peak)) break;
            else peak = next;
        }
    } else if (higherBound != 0 && peak == higherBound - 1) {                                             // :333-345
        for (;;) {
            const size_t next = peak + 1;
            if (next == N) break;
            else if (left(next) < left(peak)) break;
            else peak = next;
        }
    }
    const size_t peakOffset = peak;
    const bool offsetIsEnd = peakOffset == size_t(p.cfg.axis_points) - 1;
    // mapFrequency returns T = float: the difference is a float subtraction (TransformConstant.h:99-102)
    const size_t hi = offsetIsEnd ? peakOffset : peakOffset + 1, lo = offsetIsEnd ? (peakOffset == 0 ? 0 : peakOffset - 1) : peakOffset;
    double peakDeviance = double(p.mapped[hi] - p.mapped[lo]);
    if (p.cfg.algorithm == SGZ_ALGO_FFT && p.cfg.bin_interp != SGZ_INTERP_LANCZOS)
        peakDeviance = std::max(peakDeviance, 0.5 * double(p.N) / double(N));                             // :355-358
    out->peak_offset = double(peakOffset);
    out->peak_frequency = double(p.mapped[peakOffset]);
    out->peak_deviance = peakDeviance;
    out->peak_fraction_y = double(left(peakOffset));
    out->peak_dbs = p.cfg.low_db + double(left(peakOffset)) * (p.cfg.high_db - p.cfg.low_db);
    out->peak_slope = double(p.slope[peakOffset]);
    return SGZ_OK;
}

}  // namespace sgz

struct sgz_plan { Plan impl; };

extern "C" sgz_status sgz_track_peak_lines(const sgz_plan *plan, const float *results, double mouse_fraction, sgz_line_peak *out)
{
    if (!plan || !results || !out) return fail(SGZ_EINVAL, "null argument");
    return trackPeakLines(plan->impl, results, mouse_fraction, out);
}

extern "C" sgz_status sgz_stage_track_peak(sgz_plan *plan, const float *d_bins, double mouse_fraction, sgz_peak *out, void *stream)
{
    if (!plan || !d_bins || !out) return fail(SGZ_EINVAL, "null argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    sgz_peak *d_out = nullptr;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&d_out), sizeof(sgz_peak)));
    sgz_status st = runTrackPeak(plan->impl, d_bins, mouse_fraction, d_out, s);
    hipError_t e = hipSuccess;
    if (st == SGZ_OK) {
        e = hipMemcpyAsync(out, d_out, sizeof(sgz_peak), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    (void)hipFree(d_out);
    if (st != SGZ_OK) return st;
    if (e != hipSuccess) return hipFail(e, "sgz_stage_track_peak");
    return SGZ_OK;
}
