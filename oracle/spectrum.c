/*
 * spectrum.c -- line-by-line CPU restatement of the Signalizer Spectrum STFT chain.
 * TEST INFRASTRUCTURE (see sgz_oracle.h).  Follows, quirks included:
 *   Source/Spectrum/TransformConstant.h   (constants, frequency mapping, slope map)
 *   Source/Spectrum/TransformDSP.inl      (prepareTransform, mapToLinearSpace, DFT filters)
 *   Source/Spectrum/SpectrumDSP.cpp       (gradient colour map + additive blend + uint8)
 *   Source/Spectrum/Spectrum.cpp:226-246  (colour ratios)
 *   JuceLibraryCode/modules/juce_graphics/colour/juce_Colour.cpp (HSB round trip)
 * Phase mode (TransformDSP.inl:643-853, :1393-1432) is restated with its quirks (Q3, Q6, Q7); where the reference reads
 * working memory it never wrote (the last pixel's cancellation of a fully interpolated view) the value is defined as 0.
 */
#include "sgz_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * TransformConstant::remapFrequencies, Source/Spectrum/TransformConstant.h:125-180 */
void sgzo_remap_frequencies(const sgzo_spectrum_params *p, float *mapped)
{
    const size_t P = p->axis_points;
    const double viewSize = p->view_right - p->view_left;              /* viewRect.dist() */
    const double sampleRate = (double)p->sample_rate;
    if (p->view_scaling == SGZO_VIEW_LINEAR) {
        const double halfSampleRate = sampleRate * 0.5;
        const double complexFactor = p->channel_mode == SGZO_CH_COMPLEX ? 2.0 : 1.0;
        const double freqPerPixel = halfSampleRate / (double)(P - 1);
        for (size_t i = 0; i < P; ++i)
            mapped[i] = (float)(complexFactor * p->view_left * halfSampleRate
                                + complexFactor * viewSize * (double)i * freqPerPixel);
    } else {
        const double sampleSize = (double)(P - 1);
        const double minFreq = p->min_log_freq;
        const double end = (double)(p->sample_rate / 2);              /* T(sampleRate)/2 then double */
        if (p->channel_mode != SGZO_CH_COMPLEX) {
            for (size_t i = 0; i < P; ++i)
                mapped[i] = (float)(minFreq * pow(end / minFreq, p->view_left + viewSize * ((double)i / sampleSize)));
        } else {
            for (size_t i = 0; i < P; ++i) {
                double arg = p->view_left + viewSize * (double)i / sampleSize;
                if (arg < 0.5) {
                    mapped[i] = (float)(minFreq * pow(end / minFreq, arg * 2));
                } else {
                    arg -= 0.5;
                    const double power = minFreq * pow(end / minFreq, 1.0 - arg * 2);
                    mapped[i] = (float)(end + (end - power));
                }
            }
        }
    }
}

/* TransformConstant::generateSlopeMap, TransformConstant.h:109-118 */
void sgzo_slope_map(const sgzo_spectrum_params *p, const float *mapped, float *slope)
{
    const float a = (float)p->slope_a, b = (float)p->slope_b;
    for (size_t i = 0; i < p->axis_points; ++i)
        slope[i] = b * powf(mapped[i], a);                           /* std::pow(float, float) is powf */
}

/* Spectrum::calculateSpectrumColourRatios, Source/Spectrum/Spectrum.cpp:226-246 */
void sgzo_colour_ratios(const double ratios[SGZO_NUM_SPEC_COLOURS], float out[SGZO_NUM_SPEC_COLOURS + 1])
{
    double acc = 0.0, vals[SGZO_NUM_SPEC_COLOURS];
    for (int i = 0; i < SGZO_NUM_SPEC_COLOURS; ++i) {
        vals[i] = ratios[i] > 0.0001 ? ratios[i] : 0.0001;
        acc += vals[i];
    }
    acc += (double)FLT_EPSILON;
    out[0] = 0;
    for (int i = 0; i < SGZO_NUM_SPEC_COLOURS; ++i) out[i + 1] = (float)(vals[i] / acc);
}

/* juce::Colour::withRotatedHue via ColourHelpers::HSB,
 * JuceLibraryCode/modules/juce_graphics/colour/juce_Colour.cpp:33-107,:331-336. */
static int round_to_int(float v) { return (int)lrint((double)v); }   /* juce::roundToInt: round-half-even */
void sgzo_rotate_hue_rgb8(const uint8_t rgb[3], float amount, uint8_t out[3])
{
    const int r = rgb[0], g = rgb[1], b = rgb[2];
    const int hi = r > g ? (r > b ? r : b) : (g > b ? g : b);
    const int lo = r < g ? (r < b ? r : b) : (g < b ? g : b);
    float hue = 0, saturation = 0, brightness;
    if (hi != 0) {
        saturation = (float)(hi - lo) / (float)hi;
        if (saturation > 0) {
            const float invDiff = 1.0f / (float)(hi - lo);
            const float red = (float)(hi - r) * invDiff;
            const float green = (float)(hi - g) * invDiff;
            const float blue = (float)(hi - b) * invDiff;
            if (r == hi) hue = blue - green;
            else if (g == hi) hue = 2.0f + red - blue;
            else hue = 4.0f + green - red;
            hue *= 1.0f / 6.0f;
            if (hue < 0) ++hue;
        } else hue = 0;
    } else saturation = hue = 0;
    brightness = (float)hi / 255.0f;

    hue += amount;                                   /* withRotatedHue */

    float h = hue, s = saturation, v = brightness;   /* HSB::toRGB */
    v = v * 255.0f; if (v < 0.0f) v = 0.0f; if (v > 255.0f) v = 255.0f;
    const uint8_t intV = (uint8_t)round_to_int(v);
    if (s <= 0) { out[0] = out[1] = out[2] = intV; return; }
    if (s > 1.0f) s = 1.0f;
    h = (h - floorf(h)) * 6.0f + 0.00001f;
    const float f = h - floorf(h);
    const uint8_t x = (uint8_t)round_to_int(v * (1.0f - s));
    if (h < 1.0f)      { out[0] = intV; out[1] = (uint8_t)round_to_int(v * (1.0f - (s * (1.0f - f)))); out[2] = x; }
    else if (h < 2.0f) { out[0] = (uint8_t)round_to_int(v * (1.0f - s * f)); out[1] = intV; out[2] = x; }
    else if (h < 3.0f) { out[0] = x; out[1] = intV; out[2] = (uint8_t)round_to_int(v * (1.0f - (s * (1.0f - f)))); }
    else if (h < 4.0f) { out[0] = x; out[1] = (uint8_t)round_to_int(v * (1.0f - s * f)); out[2] = intV; }
    else if (h < 5.0f) { out[0] = (uint8_t)round_to_int(v * (1.0f - (s * (1.0f - f)))); out[1] = x; out[2] = intV; }
    else               { out[0] = intV; out[1] = x; out[2] = (uint8_t)round_to_int(v * (1.0f - s * f)); }
}

/* TransformConstant::generateSpectrogramColourRotation, TransformConstant.h:55-65 with
 * ColourRotation::operator[] (CommonSignalizer.h:931-937: withRotatedHue(index / size), float division,
 * stereo=false for the spectrogram stops, Spectrum.cpp:396-403) and FloatColour (:1139-1163). */
void sgzo_colour_table(const sgzo_spectrum_params *p, uint32_t pair, float sca[SGZO_NUM_SPEC_COLOURS + 1][3])
{
    const float size = (float)p->num_pairs;
    for (int i = 0; i <= SGZO_NUM_SPEC_COLOURS; ++i) {
        const uint32_t rotation = (i == 0) ? 0u : pair;       /* sca[0] = colourSpecs[0][0] */
        uint8_t rgb[3];
        sgzo_rotate_hue_rgb8(p->colours[i], (float)rotation / size, rgb);
        for (int c = 0; c < 3; ++c) sca[i][c] = (float)rgb[c] / 255.0f;
    }
}

/* ---------------------------------------------------------------------------------------------
 * TransformPair<T>::prepareTransform, Source/Spectrum/TransformDSP.inl:39-231.
 * The two-segment ring walk collapses to "the W newest samples in time order". */
void sgzo_prepare_transform(uint32_t mode, const float *L, const float *R, const float *w,
                            uint32_t W, uint32_t N, sgzo_cf *buf)
{
    uint32_t i;
    switch (mode) {
    case SGZO_CH_LEFT:  for (i = 0; i < W; ++i) { buf[i].re = L[i] * w[i]; buf[i].im = 0; } break;
    case SGZO_CH_RIGHT: for (i = 0; i < W; ++i) { buf[i].re = R[i] * w[i]; buf[i].im = 0; } break;
    case SGZO_CH_MERGE: for (i = 0; i < W; ++i) { buf[i].re = (L[i] + R[i]) * w[i] * 0.5f; buf[i].im = 0; } break;
    case SGZO_CH_SIDE:  for (i = 0; i < W; ++i) { buf[i].re = (L[i] - R[i]) * w[i] * 0.5f; buf[i].im = 0; } break;
    case SGZO_CH_MIDSIDE:
        for (i = 0; i < W; ++i) {
            buf[i].re = (L[i] + R[i]) * w[i] * 0.5f;
            buf[i].im = (L[i] - R[i]) * w[i] * 0.5f;
        }
        break;
    default: /* Phase, Separate, Complex */
        for (i = 0; i < W; ++i) { buf[i].re = L[i] * w[i]; buf[i].im = R[i] * w[i]; }
        break;
    }
    for (i = W; i < N; ++i) { buf[i].re = 0; buf[i].im = 0; }          /* zero-pad, :220-223 */
}

static float cf_abs(sgzo_cf z) { return hypotf(z.re, z.im); }           /* std::abs(std::complex<float>) */
static float cf_square(sgzo_cf z) { return z.re * z.re + z.im * z.im; } /* cpl::Math::square(complex) == |z|^2 */
static sgzo_cf cf_scale(float s, sgzo_cf z) { sgzo_cf r = { s * z.re, s * z.im }; return r; }
static size_t confine(size_t v, size_t lo, size_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* interpolate one pixel, shared by the three interpolation loops */
static sgzo_cf interp_at(uint32_t interp, const sgzo_cf *csf, size_t size, float pos, size_t clamp_hi)
{
    if (interp == SGZO_INTERP_LINEAR) return sgzo_linear_filter(csf, size, pos);
    if (interp == SGZO_INTERP_LANCZOS) return sgzo_lanczos_filter_wrap(csf, size, pos, 5);
    /* None: +0.5 "to centerly space bins", TransformDSP.inl:577 (double add) */
    return csf[confine((size_t)((double)pos + 0.5), 0, clamp_hi)];
}

/* TransformPair<T>::mapToLinearSpace, FFT branch, Source/Spectrum/TransformDSP.inl:506-1102. */
int sgzo_map_to_linear_space(const sgzo_spectrum_params *p, const float *mf, double window_scale,
                             sgzo_cf *csf, uint32_t Nu, sgzo_cf *csp)
{
    const long N = (long)Nu;
    const long P = (long)p->axis_points;
    if (N < 3 || p->sample_rate < 1) return 0;                            /* :519 */
    const size_t numBins = (size_t)(N >> 1);
    const float topFrequency = p->sample_rate / 2;                         /* :523 */
    const float freqToBin = (float)((float)numBins / topFrequency);        /* :524 */
    const float invSize = (float)(window_scale / ((double)p->window_size * 0.5)); /* :540 */
    const size_t csfSize = (size_t)N + 1;
    long x = 0, bin = 0, oldBin = 0, maxLBin, maxRBin = 0;
    float maxLMag, maxRMag, newLMag, newRMag;

    switch (p->channel_mode) {
    case SGZO_CH_LEFT: case SGZO_CH_RIGHT: case SGZO_CH_MERGE: case SGZO_CH_SIDE: {  /* :544-642 */
        csf[0] = cf_scale(0.5f, csf[0]);
        csf[N >> 1] = cf_scale(0.5f, csf[N >> 1]);
        for (size_t i = 0; i < numBins; ++i) { csf[i].re = cf_abs(csf[i]); csf[i].im = 0; }
        const double fftBandwidth = 1.0 / (double)numBins;
        for (x = 0; x < P - 1; ++x) {
            const double bwForLine = (double)((mf[x + 1] - mf[x]) / topFrequency);
            if (bwForLine > fftBandwidth) break;
            csp[x] = cf_scale(invSize, interp_at(p->bin_interp, csf, csfSize, mf[x] * freqToBin, numBins));
        }
        oldBin = (long)(mf[x] * freqToBin);
        for (; x < P; ++x) {
            maxLMag = newLMag = 0;
            bin = (long)(size_t)(mf[x] * freqToBin);
            maxLBin = bin;
            long diff = bin - oldBin;
            long counter = diff ? 1 : 0;
            do {
                const long offset = oldBin + counter;
                newLMag = cf_square(csf[offset]);
                if (newLMag > maxLMag) { maxLBin = oldBin + counter; maxLMag = newLMag; }
                counter++; diff--;
            } while (diff > 0);
            csp[x] = cf_scale(invSize, csf[maxLBin]);
            oldBin = bin;
        }
        return 1;
    }
    case SGZO_CH_SEPARATE: case SGZO_CH_MIDSIDE: {                                    /* :854-986 */
        sgzo_separate_transforms_ipl(csf, (uint32_t)N);
        csf[N].re = csf[0].im * 0.5f; csf[N].im = 0;                                  /* :861 */
        csf[0].re = csf[0].re * 0.5f; csf[0].im = 0;                                  /* :862 */
        csf[N >> 1] = cf_scale(0.5f, csf[N >> 1]);
        csf[(N >> 1) - 1] = cf_scale(0.5f, csf[(N >> 1) - 1]);                       /* quirk Q3 */
        for (long i = 1; i < N; ++i) { csf[i].re = cf_abs(csf[i]); csf[i].im = 0; }
        const double fftBandwidth = 1.0 / (double)numBins;
        for (x = 0; x < P - 1; ++x) {
            const double bwForLine = (double)((mf[x + 1] - mf[x]) / topFrequency);
            if (bwForLine > fftBandwidth) break;
            const float pos = mf[x] * freqToBin;
            if (p->bin_interp == SGZO_INTERP_NONE) {
                const size_t index = confine((size_t)((double)pos + 0.5), 0, numBins);
                csp[x] = cf_scale(invSize, csf[index]);
                csp[P + x] = cf_scale(invSize, csf[(size_t)N - index]);
            } else {
                /* N - (float) : int -> float arithmetic, :893,:912 */
                const float rpos = (float)N - pos;
                csp[x] = cf_scale(invSize, interp_at(p->bin_interp, csf, csfSize, pos, numBins));
                csp[P + x] = cf_scale(invSize, interp_at(p->bin_interp, csf, csfSize, rpos, numBins));
            }
        }
        oldBin = (long)(mf[x] * freqToBin);
        for (; x < P; ++x) {
            maxLMag = maxRMag = newLMag = newRMag = 0;
            bin = (long)(size_t)(mf[x] * freqToBin);
            maxRBin = maxLBin = bin;
            long diff = bin - oldBin;
            long counter = diff ? 1 : 0;
            do {
                const long offset = oldBin + counter;
                newLMag = cf_square(csf[offset]);
                newRMag = cf_square(csf[N - offset]);
                if (newLMag > maxLMag) { maxLBin = oldBin + counter; maxLMag = newLMag; }
                if (newRMag > maxRMag) { maxRBin = N - (oldBin + counter); maxRMag = newRMag; }
                counter++; diff--;
            } while (diff > 0);
            csp[x] = cf_scale(invSize, csf[maxLBin]);
            csp[P + x] = cf_scale(invSize, csf[maxRBin]);
            oldBin = bin;
        }
        return 1;
    }
    case SGZO_CH_COMPLEX: {                                                            /* :987-1097 */
        csf[0] = cf_scale(0.5f, csf[0]);
        const double fftBandwidth = 1.0 / (double)(numBins * 2);
        for (long i = 1; i < N; ++i) { csf[i].re = cf_abs(csf[i]); csf[i].im = 0; }
        x = 0;
        while (x < P) {
            for (; x < P; ++x) {
                if (x != P - 1) {
                    const double bwForLine = (double)((mf[x + 1] - mf[x]) / topFrequency);
                    if (bwForLine > fftBandwidth) break;
                }
                csp[x] = cf_scale(invSize, interp_at(p->bin_interp, csf, csfSize, mf[x] * freqToBin, (size_t)N));
            }
            if (x != P) oldBin = (long)(mf[x] * freqToBin);
            for (; x < P; ++x) {
                maxLMag = newLMag = 0;
                bin = (long)(size_t)(mf[x] * freqToBin);
                maxLBin = bin;
                if (x != P - 1) {
                    const double bwForLine = (double)((mf[x + 1] - mf[x]) / topFrequency);
                    if (bwForLine < fftBandwidth) break;
                }
                long diff = bin - oldBin;
                long counter = diff ? 1 : 0;
                do {
                    const long offset = oldBin + counter;
                    newLMag = cf_square(csf[offset]);
                    if (newLMag > maxLMag) { maxLBin = oldBin + counter; maxLMag = newLMag; }
                    counter++; diff--;
                } while (diff > 0);
                csp[x] = cf_scale(invSize, csf[maxLBin]);
                oldBin = bin;
            }
        }
        return 1;
    }
    case SGZO_CH_PHASE: {                                                              /* :643-853 */
        /* wsp[2x] = magnitude, wsp[2x+1] = phase cancellation: the same working memory as csp, viewed as floats (:508-509) */
        float *wsp = (float *)csp;
        sgzo_separate_transforms_ipl(csf, (uint32_t)N);
        csf[N].re = csf[0].im * 0.5f; csf[N].im = 0;                                  /* :649 */
        csf[0].re = csf[0].re * 0.5f; csf[0].im = 0;                                  /* :650 */
        csf[N >> 1] = cf_scale(0.5f, csf[N >> 1]);                                   /* :651 */
        csf[(N >> 1) - 1] = cf_scale(0.5f, csf[(N >> 1) - 1]);                       /* :652, quirk Q3 */
        size_t bandWidthBreakingPoint = (size_t)P;                                    /* :665 */
        const double fftBandwidth = 1.0 / (double)numBins;
        if (p->bin_interp == SGZO_INTERP_LINEAR || p->bin_interp == SGZO_INTERP_LANCZOS) {
            const size_t filterSize = p->bin_interp == SGZO_INTERP_LINEAR ? 1 : 5;    /* linearFilterSize :712 / lanczosFilterSize */
            /* phase pass (:674-690 / :732-750): on the un-normalised vectors.  The last pixel's cancellation is never
             * written when the loop does not break (stale working memory in the reference; defined as 0 here). */
            wsp[(P - 1) * 2 + 1] = 0;
            for (x = 0; x < P - 1; ++x) {
                const double bwForLine = (double)((mf[x + 1] - mf[x]) / topFrequency);
                if (bwForLine > fftBandwidth) { bandWidthBreakingPoint = (size_t)x; break; }
                const float pos = mf[x] * freqToBin;
                const sgzo_cf iLeft = interp_at(p->bin_interp, csf, csfSize, pos, numBins);
                const sgzo_cf iRight = interp_at(p->bin_interp, csf, csfSize, (float)N - pos, numBins);
                sgzo_cf sum; sum.re = iLeft.re + iRight.re; sum.im = iLeft.im + iRight.im;
                const float cancellation = invSize * sqrtf(cf_square(sum));
                const float mid = invSize * (cf_abs(iLeft) + cf_abs(iRight));
                wsp[x * 2 + 1] = 1.0f - (mid > 0 ? (cancellation / mid) : 0);
            }
            /* magnitude pass (:707-727 / :759-775): bins are normalised (replaced by their magnitude) lazily, just ahead of
             * the filter window, and only while x < breakingPoint - (filterSize + 1) -- an unsigned subtraction (Q6) */
            size_t normalizedPosition = 0;
            for (x = 0; x < (long)bandWidthBreakingPoint; ++x) {
                const float binPosition = mf[x] * freqToBin;
                while ((binPosition + (float)filterSize) > (float)normalizedPosition &&
                       (size_t)x < bandWidthBreakingPoint - (filterSize + 1)) {
                    csf[normalizedPosition].re = cf_abs(csf[normalizedPosition]); csf[normalizedPosition].im = 0;
                    csf[(size_t)N - normalizedPosition].re = cf_abs(csf[(size_t)N - normalizedPosition]);
                    csf[(size_t)N - normalizedPosition].im = 0;
                    normalizedPosition++;
                }
                const sgzo_cf iLeft = interp_at(p->bin_interp, csf, csfSize, binPosition, numBins);
                const sgzo_cf iRight = interp_at(p->bin_interp, csf, csfSize, (float)N - binPosition, numBins);
                wsp[x * 2] = invSize * (cf_abs(iLeft) + cf_abs(iRight));
            }
        } else {                                                                      /* None, :779-801 */
            for (x = 0; x < P - 1; ++x) {
                const double bwForLine = (double)((mf[x + 1] - mf[x]) / topFrequency);
                if (bwForLine > fftBandwidth) break;
                const size_t index = confine((size_t)((double)(mf[x] * freqToBin) + 0.5), 0, numBins - 1);
                const sgzo_cf iLeft = csf[index], iRight = csf[(size_t)N - index];
                sgzo_cf sum; sum.re = iLeft.re + iRight.re; sum.im = iLeft.im + iRight.im;
                const float cancellation = invSize * cf_abs(sum);
                const float mid = invSize * (cf_abs(iLeft) + cf_abs(iRight));
                wsp[x * 2] = mid;
                wsp[x * 2 + 1] = 1.0f - (mid > 0 ? (cancellation / mid) : 0);
            }
        }
        if (x < P) oldBin = (long)(mf[x] * freqToBin);                                /* :806-807 */
        for (; x < P; ++x) {                                                          /* :811-849 */
            size_t maxBin = 0;
            float maxValue = 0, newMag = 0;
            bin = (long)(size_t)(mf[x] * freqToBin);
            long diff = bin - oldBin;
            long counter = diff ? 1 : 0;
            do {
                const long offset = oldBin + counter;
                const float a = cf_square(csf[offset]), b = cf_square(csf[N - offset]);
                newMag = a < b ? b : a;                                               /* std::max */
                if (newMag > maxValue) { maxValue = newMag; maxBin = (size_t)(oldBin + counter); }
                counter++; diff--;
            } while (diff > 0);
            const sgzo_cf leftMax = csf[maxBin], rightMax = csf[(size_t)N - maxBin];
            sgzo_cf sum; sum.re = leftMax.re + rightMax.re; sum.im = leftMax.im + rightMax.im;
            const float interference = invSize * cf_abs(sum);
            const float mid = invSize * (cf_abs(leftMax) + cf_abs(rightMax));
            const float cancellation = interference / mid;
            wsp[x * 2] = mid;
            wsp[x * 2 + 1] = 1.0f - (mid > 0 ? cancellation : 0);
            oldBin = bin;
        }
        return 1;
    }
    default:
        return -1;
    }
}

static int state_channels(uint32_t mode) { return mode > SGZO_CH_SIDE ? 2 : 1; }  /* TransformConstant.h:183-186 */

/* TransformPair<T>::mapAndTransformDFTFilters, Source/Spectrum/TransformDSP.inl:1299-1435.
 * states/results layout: [graph k][pixel i] as UComplexFilter {magnitude|leftMagnitude, phase|rightMagnitude}. */
void sgzo_map_and_transform_filters(const sgzo_spectrum_params *p, const float *slope,
                                    const sgzo_cf *csp, sgzo_cf *states, sgzo_cf *results)
{
    const size_t size = p->axis_points;
    const double lowerFraction = pow(10.0, p->low_db / 20.0);     /* cpl::Math::dbToFraction<double> */
    const double upperFraction = pow(10.0, p->high_db / 20.0);
    const float deltaYRecip = (float)(1.0 / log(upperFraction / lowerFraction));
    const float minFracRecip = (float)(1.0 / lowerFraction);
    const float lowerClip = (float)p->clip_db;
    const float *newVals = (const float *)csp;

    switch (p->channel_mode) {
    case SGZO_CH_LEFT: case SGZO_CH_MERGE: case SGZO_CH_RIGHT: case SGZO_CH_SIDE: case SGZO_CH_COMPLEX:
        for (size_t i = 0; i < size; ++i) {
            const float newReal = newVals[i * 2], newImag = newVals[i * 2 + 1];
            const float magnitude = sqrtf(newReal * newReal + newImag * newImag);
            for (int k = 0; k < SGZO_NUM_GRAPHS; ++k) {
                sgzo_cf *st = &states[(size_t)k * size + i], *rs = &results[(size_t)k * size + i];
                st->re *= p->pole[k];
                if (magnitude > st->re) st->re = magnitude;
                const float deltaX = slope[i] * st->re * minFracRecip;
                rs->re = deltaX > 0 ? logf(deltaX) * deltaYRecip : lowerClip;
                rs->im = 0;
            }
        }
        break;
    case SGZO_CH_SEPARATE: case SGZO_CH_MIDSIDE:
        for (size_t i = 0; i < size; ++i) {
            const float lreal = newVals[i * 2], rreal = newVals[i * 2 + size * 2];
            const float limag = newVals[i * 2 + 1], rimag = newVals[i * 2 + size * 2 + 1];
            const float lmag = sqrtf(lreal * lreal + limag * limag);
            const float rmag = sqrtf(rreal * rreal + rimag * rimag);
            for (int k = 0; k < SGZO_NUM_GRAPHS; ++k) {
                sgzo_cf *st = &states[(size_t)k * size + i], *rs = &results[(size_t)k * size + i];
                st->re *= p->pole[k];
                st->im *= p->pole[k];
                if (lmag > st->re) st->re = lmag;
                if (rmag > st->im) st->im = rmag;
                const float deltaLX = slope[i] * st->re * minFracRecip;
                const float deltaRX = slope[i] * st->im * minFracRecip;
                rs->re = deltaLX > 0 ? logf(deltaLX) * deltaYRecip : lowerClip;
                rs->im = deltaRX > 0 ? logf(deltaRX) * deltaYRecip : lowerClip;
            }
        }
        break;
    case SGZO_CH_PHASE: {                                                              /* :1393-1432 */
        /* std::pow<T>(pole, 0.3): restated as powf (UNVERIFIED: the explicit template argument selects an overload we cannot see) */
        float phaseFilters[SGZO_NUM_GRAPHS];
        for (int k = 0; k < SGZO_NUM_GRAPHS; ++k) phaseFilters[k] = powf(p->pole[k], 0.3f);
        for (size_t i = 0; i < size; ++i) {
            float mag = newVals[i * 2];
            float phase = newVals[i * 2 + 1];
            mag *= 0.5f;
            for (int k = 0; k < SGZO_NUM_GRAPHS; ++k) {
                sgzo_cf *st = &states[(size_t)k * size + i], *rs = &results[(size_t)k * size + i];
                st->re *= p->pole[k];
                if (mag > st->re) st->re = mag;
                phase *= mag;                                                          /* inside the graph loop: quirk Q7 */
                st->im = phase + phaseFilters[k] * (st->im - phase);
                const float deltaX = slope[i] * st->re * minFracRecip;
                const float deltaY = slope[i] * st->im * minFracRecip;
                rs->re = deltaX > 0 ? logf(deltaX) * deltaYRecip : lowerClip;
                rs->im = deltaY > 0 ? logf(deltaY) * deltaYRecip : lowerClip;
            }
        }
        break;
    }
    default: break;
    }
}

/* AudioDispatcher::blendAndDispatchSpectrums, Source/Spectrum/SpectrumDSP.cpp:111-206.
 * frames: [pair][P]; only .magnitude (== leftMagnitude) of LineMain colours the column (:123). */
void sgzo_blend_column(const sgzo_spectrum_params *p, const float *ratios /*6*/,
                       const sgzo_cf *frames, uint32_t num_pairs, uint8_t *rgba)
{
    const size_t P = p->axis_points;
    float (*colourBuffer)[3] = (float (*)[3])calloc(P, sizeof(float[3]));
    for (uint32_t pr = 0; pr < num_pairs; ++pr) {
        float sca[SGZO_NUM_SPEC_COLOURS + 1][3];
        sgzo_colour_table(p, pr, sca);
        const sgzo_cf *input = frames + (size_t)pr * P;
        for (size_t i = 0; i < P; ++i) {
            const float intensity = input[i].re;
            if (intensity < 0) continue;
            float colour[3];
            /* the reference leaves `colour` uninitialised if no stop matches (practically unreachable,
             * sum(ratios) = 1 - eps > 0.999); the oracle defines that case as the last stop. */
            colour[0] = sca[SGZO_NUM_SPEC_COLOURS][0];
            colour[1] = sca[SGZO_NUM_SPEC_COLOURS][1];
            colour[2] = sca[SGZO_NUM_SPEC_COLOURS][2];
            if (intensity < 0.999f) {
                float accumulatedSum = 0;
                for (int c = 1; c <= SGZO_NUM_SPEC_COLOURS; ++c) {
                    const float nextScale = ratios[c];
                    accumulatedSum += nextScale;
                    if (accumulatedSum >= intensity) {
                        const float min = accumulatedSum - nextScale;
                        const float max = accumulatedSum;
                        const float mix = (intensity - min) / (max - min);
                        const float imix = 1 - mix;
                        const float *a = sca[c - 1], *b = sca[c];
                        colour[0] = a[0] * imix + b[0] * mix;
                        colour[1] = a[1] * imix + b[1] * mix;
                        colour[2] = a[2] * imix + b[2] * mix;
                        break;
                    }
                }
            }
            for (int c = 0; c < 3; ++c)
                colourBuffer[i][c] += (1 - colourBuffer[i][c]) * colour[c];    /* GL_ONE_MINUS_SRC_COLOR */
        }
    }
    for (size_t i = 0; i < P; ++i) {
        rgba[i * 4 + 0] = (uint8_t)(colourBuffer[i][0] * 255);
        rgba[i * 4 + 1] = (uint8_t)(colourBuffer[i][1] * 255);
        rgba[i * 4 + 2] = (uint8_t)(colourBuffer[i][2] * 255);
        rgba[i * 4 + 3] = 255;
    }
    free(colourBuffer);
}

long sgzo_num_frames(size_t nsamples, uint32_t W, uint32_t hop)
{
    if (nsamples < W || hop == 0) return 0;
    return (long)((nsamples - W) / hop) + 1;
}

/* Offline job: ideal STFT framing (frame f covers samples [f*hop, f*hop+W)), i.e. what
 * TransformPair::audioEntryPoint (TransformDSP.inl:1165-1211) produces when every frame fires at a
 * callback end with history == W, followed by blendAndDispatchSpectrums per frame. */
static void fft_restated(sgzo_cf *buf, uint32_t N) { sgzo_fft_forward(buf, N); }
static void fft_vectorised(sgzo_cf *buf, uint32_t N) { if (sgzo_fft_forward_simd(buf, N)) sgzo_fft_forward(buf, N); }

static long spectrogram_impl(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                             long f0, long f1, uint8_t *rgba_out, sgzo_cf *line_out, sgzo_cf *mapped_out,
                             void (*fft)(sgzo_cf *, uint32_t))
{
    const uint32_t W = p->window_size, N = sgzo_transform_size(W), P = p->axis_points, C = p->num_pairs;
    const long F = sgzo_num_frames(nsamples, W, p->hop);
    if (F <= 0 || P < 2) return 0;
    if (f1 > F) f1 = F;
    float *window = (float *)calloc(N, sizeof(float));
    float *mapped = (float *)malloc(sizeof(float) * P);
    float *slope = (float *)malloc(sizeof(float) * P);
    float ratios[SGZO_NUM_SPEC_COLOURS + 1];
    const double scale = sgzo_window(p->window_type, p->window_symmetry, p->window_alpha, p->window_beta, W, window);
    sgzo_remap_frequencies(p, mapped);
    sgzo_slope_map(p, mapped, slope);
    sgzo_colour_ratios(p->ratios, ratios);
    sgzo_cf *csf = (sgzo_cf *)calloc((size_t)N + 1, sizeof(sgzo_cf));
    sgzo_cf *csp = (sgzo_cf *)calloc((size_t)P * 2, sizeof(sgzo_cf));
    sgzo_cf *states = (sgzo_cf *)calloc((size_t)C * SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
    sgzo_cf *results = (sgzo_cf *)calloc((size_t)C * SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
    sgzo_cf *frames = (sgzo_cf *)calloc((size_t)C * P, sizeof(sgzo_cf));
    const int sc = state_channels(p->channel_mode);

    for (long f = f0; f < f1; ++f) {
        const size_t start = (size_t)f * p->hop;
        for (uint32_t pr = 0; pr < C; ++pr) {
            const float *L = planar[2 * pr] + start, *R = planar[2 * pr + 1] + start;
            sgzo_prepare_transform(p->channel_mode, L, R, window, W, N, csf);
            csf[N].re = csf[N].im = 0;                  /* mono modes never write csf[N]; defined as 0 */
            fft(csf, N);
            memset(csp, 0, sizeof(sgzo_cf) * (size_t)P * 2);
            sgzo_map_to_linear_space(p, mapped, scale, csf, N, csp);
            if (mapped_out)
                memcpy(mapped_out + ((size_t)(f - f0) * C + pr) * 2 * P, csp, sizeof(sgzo_cf) * (size_t)P * sc);
            sgzo_cf *st = states + (size_t)pr * SGZO_NUM_GRAPHS * P;
            sgzo_cf *rs = results + (size_t)pr * SGZO_NUM_GRAPHS * P;
            sgzo_map_and_transform_filters(p, slope, csp, st, rs);
            memcpy(frames + (size_t)pr * P, rs, sizeof(sgzo_cf) * P);            /* addAudioFrame :1144-1147 */
            if (line_out)
                memcpy(line_out + ((size_t)(f - f0) * C + pr) * SGZO_NUM_GRAPHS * P, rs,
                       sizeof(sgzo_cf) * SGZO_NUM_GRAPHS * P);
        }
        if (rgba_out) sgzo_blend_column(p, ratios, frames, C, rgba_out + (size_t)(f - f0) * P * 4);
    }
    free(window); free(mapped); free(slope); free(csf); free(csp); free(states); free(results); free(frames);
    return f1 - f0;
}

long sgzo_spectrogram(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                      uint8_t *rgba_out, sgzo_cf *line_out, sgzo_cf *mapped_out)
{
    return spectrogram_impl(p, planar, nsamples, 0, sgzo_num_frames(nsamples, p->window_size, p->hop),
                            rgba_out, line_out, mapped_out, fft_restated);
}

long sgzo_spectrogram_range(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                            long f0, long f1, uint8_t *rgba_out)
{
    return spectrogram_impl(p, planar, nsamples, f0, f1, rgba_out, NULL, NULL, fft_restated);
}

/* bench.py's cpu_baseline.simd_value ONLY: the same chain with fft_simd.c's vectorisable transform in the place of the restated
 * radix-2 (the reference's is pffft, SIMD: TransformDSP.inl:487-502).  Not the parity oracle: nothing is compared against it. */
long sgzo_spectrogram_range_simd(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                                 long f0, long f1, uint8_t *rgba_out)
{
    return spectrogram_impl(p, planar, nsamples, f0, f1, rgba_out, NULL, NULL, fft_vectorised);
}

/* Test hook: mapAndTransformDFTFilters + blendAndDispatchSpectrums (the two functions above, unchanged) over F frames of
 * GIVEN csp values [F][C][2P] -- the decay / dB / colour stages on somebody else's mapped pixels, states starting from zero.
 * This is what makes the end-to-end parity argument a chain: bins within the FFT tolerance, mapping bit-exact given bins,
 * colour bit-exact given mapped pixels. */
long sgzo_decay_colour(const sgzo_spectrum_params *p, const sgzo_cf *csp_all, long F, uint8_t *rgba_out, sgzo_cf *line_out)
{
    const uint32_t P = p->axis_points, C = p->num_pairs;
    if (F <= 0 || P < 2) return 0;
    float *mapped = (float *)malloc(sizeof(float) * P);
    float *slope = (float *)malloc(sizeof(float) * P);
    float ratios[SGZO_NUM_SPEC_COLOURS + 1];
    sgzo_remap_frequencies(p, mapped);
    sgzo_slope_map(p, mapped, slope);
    sgzo_colour_ratios(p->ratios, ratios);
    sgzo_cf *states = (sgzo_cf *)calloc((size_t)C * SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
    sgzo_cf *results = (sgzo_cf *)calloc((size_t)C * SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
    sgzo_cf *frames = (sgzo_cf *)calloc((size_t)C * P, sizeof(sgzo_cf));
    for (long f = 0; f < F; ++f) {
        for (uint32_t pr = 0; pr < C; ++pr) {
            sgzo_cf *st = states + (size_t)pr * SGZO_NUM_GRAPHS * P;
            sgzo_cf *rs = results + (size_t)pr * SGZO_NUM_GRAPHS * P;
            sgzo_map_and_transform_filters(p, slope, csp_all + ((size_t)f * C + pr) * 2 * P, st, rs);
            memcpy(frames + (size_t)pr * P, rs, sizeof(sgzo_cf) * P);
            if (line_out)
                memcpy(line_out + ((size_t)f * C + pr) * SGZO_NUM_GRAPHS * P, rs, sizeof(sgzo_cf) * SGZO_NUM_GRAPHS * P);
        }
        if (rgba_out) sgzo_blend_column(p, ratios, frames, C, rgba_out + (size_t)f * P * 4);
    }
    free(mapped); free(slope); free(states); free(results); free(frames);
    return F;
}

/* Test hook: libm's logf over an array (the device's port of glibc's algorithm is checked against it over every float). */
void sgzo_logf_array(const float *x, float *y, size_t n)
{
    for (size_t i = 0; i < n; ++i) y[i] = logf(x[i]);
}

/* Frequency tracker, the raw-FFT branch of Spectrum::drawFrequencyTracking (Source/Spectrum/SpectrumRendering.cpp:379-469; SURVEY 8(f)
 * #4): nearest peak of |source|^2 around the mouse position (first maximum, then the walk along a still rising edge at a boundary of
 * the search range), parabolic fit through the three dB values around it (the log-domain fit of JOS's PARSHL notes, :426-444).
 * source: the transform's working memory after mapToLinearSpace (csf, N + 1 entries); mapped: mappedFrequencies[P].
 * out: {peakOffset, peakFraction, peakFrequency, peakDBs (before the slope correction), alpha, beta, gamma, phi}. */
void sgzo_track_peak(const sgzo_spectrum_params *p, const sgzo_cf *source, uint32_t Nt, const float *mapped, double window_scale,
                     double mouse_fraction, double out[8])
{
    const double nearbyFractionToConsider = 0.03;
    const double sampleRate = (double)p->sample_rate;
    const size_t N = Nt;
    const int points = (int)p->axis_points;                                            /* getNumFilters() */
    long lowerBound = (long)llround((double)points * (mouse_fraction - nearbyFractionToConsider));
    long c = lowerBound < 0 ? 0 : (lowerBound > points - 1 ? points - 1 : lowerBound);
    lowerBound = (long)llround((double)((float)N * mapped[c]) / sampleRate);           /* N * mapFrequency(..): size_t * float */
    long higherBound = (long)llround((double)points * (mouse_fraction + nearbyFractionToConsider));
    c = higherBound < 0 ? 0 : (higherBound > points - 1 ? points - 1 : higherBound);
    higherBound = (long)llround((double)((float)N * mapped[c]) / sampleRate);
    lowerBound = lowerBound < 0 ? 0 : (lowerBound > (long)N ? (long)N : lowerBound);
    higherBound = higherBound < 0 ? 0 : (higherBound > (long)N ? (long)N : higherBound);
#define SQ(z) ((z).re * (z).re + (z).im * (z).im)                                      /* cpl::Math::square(complex), UNVERIFIED vs cpl */
    long peak = lowerBound;                                                            /* std::max_element: the first largest */
    for (long k = lowerBound + 1; k <= higherBound; ++k)
        if (SQ(source[peak]) < SQ(source[k])) peak = k;
    if (peak == lowerBound && lowerBound != 0) {                                       /* :400-413 */
        for (;;) {
            const long next = peak - 1;
            if (next == 0) break;
            else if (SQ(source[next]) < SQ(source[peak])) break;
            else peak = next;
        }
    } else if (peak == higherBound - 1) {                                              /* :414-427 */
        for (;;) {
            const long next = peak + 1;
            if (next == (long)N) break;                                                /* source.end() */
            else if (SQ(source[next]) < SQ(source[peak])) break;
            else peak = next;
        }
    }
#undef SQ
    const long peakOffset = peak;
    const float invSize = (float)(window_scale / ((double)p->window_size * 0.5));
    const long ia = peakOffset == 0 ? 0 : peakOffset - 1, ic = peakOffset == (long)N ? peakOffset : peakOffset + 1;
#define ABSI(z) hypotf((z).re * invSize, (z).im * invSize)                              /* std::abs(source[k] * invSize) */
    const float alpha = 20 * log10f(ABSI(source[ia]));
    const float beta = 20 * log10f(ABSI(source[peakOffset]));
    const float gamma = 20 * log10f(ABSI(source[ic]));
#undef ABSI
    const double phi = 0.5 * (alpha - gamma) / (alpha - 2 * beta + gamma);
    const double peakFraction = 2 * ((double)peakOffset + (isnormal(phi) ? phi : 0)) / (double)N;
    const double peakFrequency = 0.5 * peakFraction * sampleRate;
    double peakDBs = beta - 0.25 * (alpha - gamma) * phi;
    if (!isnormal(peakDBs)) peakDBs = 20 * log10((double)hypotf(source[peakOffset].re, source[peakOffset].im) / ((double)N * 0.5));
    out[0] = (double)peakOffset; out[1] = peakFraction; out[2] = peakFrequency; out[3] = peakDBs;
    out[4] = alpha; out[5] = beta; out[6] = gamma; out[7] = phi;
}

/* The tracker's line-results branch, Spectrum::drawFrequencyTracking (Source/Spectrum/SpectrumRendering.cpp:300-377): Complex mode, the
 * RSNT algorithm and the LineMain / LineSecond graphs look for the peak in lineGraphs[graphN].getResults(axisPoints).
 * results: UComplex [P] as (leftMagnitude, rightMagnitude); mapped: mappedFrequencies[P]; slope: slopeMap[P].
 * out: {peakOffset, peakFrequency, peakDeviance, peakFractionY, peakDBs, peakSlope}.
 * (cpl::Math::round / confineTo / UnityScale::linear are absent: round-half-away, clamp, min + x (max - min).) */
void sgzo_track_peak_lines(const sgzo_spectrum_params *p, const float *results, const float *mapped, const float *slope, uint32_t transform_size,
                           double mouse_fraction, double out[6])
{
    double mouseFraction = mouse_fraction < 0 ? 0 : (mouse_fraction > 1 ? 1 : mouse_fraction);          /* :292 */
    const double nearbyFractionToConsider = 0.03;
    const size_t N = p->axis_points;                                                                 /* results.size() */
    const size_t pivot = (size_t)llround((double)N * mouseFraction);                                 /* :308 */
    const size_t range = (size_t)llround((double)N * nearbyFractionToConsider);
    const size_t lowerBound = range > pivot ? 0 : pivot - range;
    const size_t higherBound = range + pivot > N ? N : range + pivot;
#define LEFT(i) results[2 * (i)]
    /* std::max_element(begin + lowerBound, begin + higherBound, left.leftMagnitude < right.leftMagnitude): the first largest */
    size_t peak = lowerBound < N ? lowerBound : N - 1;                                               /* (empty range: stay inside the results) */
    for (size_t i = lowerBound + 1; i < higherBound; ++i)
        if (LEFT(peak) < LEFT(i)) peak = i;
    if (peak == lowerBound && lowerBound != 0) {                                                     /* :320-332 */
        while (1) {
            const size_t nextPeak = peak - 1;
            if (nextPeak == 0) break;                                                                /* nextPeak == results.begin() */
            else if (LEFT(nextPeak) < LEFT(peak)) break;
            else peak = nextPeak;
        }
    } else if (higherBound != 0 && peak == higherBound - 1) {                                        /* :333-345 */
        while (1) {
            const size_t nextPeak = peak + 1;
            if (nextPeak == N) break;                                                                /* results.end() */
            else if (LEFT(nextPeak) < LEFT(peak)) break;
            else peak = nextPeak;
        }
    }
    const size_t peakOffset = peak;
    const double peakFrequency = (double)mapped[peakOffset];                                         /* constant.mapFrequency(peakOffset) */
    const int offsetIsEnd = peakOffset == (size_t)p->axis_points - 1;
    const size_t hi = offsetIsEnd ? peakOffset : peakOffset + 1, lo = offsetIsEnd ? (peakOffset == 0 ? 0 : peakOffset - 1) : peakOffset;
    double peakDeviance = (double)(mapped[hi] - mapped[lo]);                                         /* T = float arithmetic */
    if (p->algorithm == 0 /* FFT */ && p->bin_interp != 2 /* Lanczos */) {
        const double alt = 0.5 * (double)transform_size / (double)N;
        if (peakDeviance < alt) peakDeviance = alt;                                                  /* std::max */
    }
    const double peakFractionY = (double)LEFT(peakOffset);
#undef LEFT
    out[0] = (double)peakOffset; out[1] = peakFrequency; out[2] = peakDeviance; out[3] = peakFractionY;
    out[4] = p->low_db + peakFractionY * (p->high_db - p->low_db);
    out[5] = (double)slope[peakOffset];
}
