/*
 * resonator.c -- oracle for the Spectrum view's second transform algorithm, SpectrumContent::TransformAlgorithm::RSNT
 * ("Resonator", Source/Spectrum/SpectrumParameters.h:68,159).  TEST INFRASTRUCTURE ONLY (see sgz_oracle.h).
 *
 * What is Signalizer's and is restated line by line (file:line at each function):
 *   - the channel dispatch of the audio thread, TransformPair::resonatingDispatch (Source/Spectrum/TransformDSP.inl:1213-1295);
 *   - the frame cadence: resonate `availableSamples`, emit a frame every sampleBufferSize samples (audioEntryPoint :1165-1211);
 *   - the RSNT branch of mapToLinearSpace (:1103-1133): copy the windowed state out, Phase post-processing;
 *   - mapAndTransformDFTFilters / blendAndDispatchSpectrums downstream (spectrum.c, unchanged).
 *
 * What is cpl's and ABSENT (cpl::dsp::CComplexResonator<T, 2>: Constant::mapSystemHz, resonateReal, getWholeWindowedState;
 * cpl::dsp::windowCoefficients; cpl::Math::square) -- External/cpl is an empty submodule.  Those are restated here from the
 * published mathematics they implement, every one of them tagged UNVERIFIED vs cpl:
 *   - a bank of complex one-pole resonators  s[n] = c s[n-1] + x[n],  c = r e^{j w}  (the recursive form of a DFT bin with an
 *     exponential window; r from the -3 dB bandwidth B in Hz: r = exp(-pi B / fs));
 *   - frequency-domain windowing: for a cosine-sum window  w[n] = sum_m (-1)^m a_m cos(2 pi m n / N)  the windowed transform is
 *     a_0 X[k] - a_1/2 (X[k-1] + X[k+1]) + a_2/2 (X[k-2] + X[k+2]) - ...  -- evaluated on resonators detuned by +-m B ("vectors");
 *   - which constants cpl uses for gain, for the bandwidth bound when Q is not free, and in which order it sums are NOT knowable
 *     here; the choices below are this build's and are stated where they are made.  Parity for this mode is therefore between the
 *     HIP path and THIS restatement only.
 */
#include "sgz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* cpl::dsp::windowCoefficients<T>(type) -> {coefficients, count} (Spectrum.cpp:593 passes `.second` as numVectors).  UNVERIFIED vs cpl:
 * the textbook cosine-sum coefficients (Harris 1978; Nuttall 1981; Heinzel 2002 for the flat top).  Windows without a cosine-sum
 * form (triangular, Welch, Gaussian, Kaiser) have no frequency-domain kernel of finite support: one term, i.e. unwindowed. */
int sgzo_window_cosine_terms(uint32_t window_type, double a[SGZO_RES_MAX_TERMS])
{
    memset(a, 0, sizeof(double) * SGZO_RES_MAX_TERMS);
    switch (window_type) {
    case SGZO_WIN_HANN: a[0] = 0.5; a[1] = 0.5; return 2;
    case SGZO_WIN_HAMMING: a[0] = 0.54; a[1] = 0.46; return 2;
    case SGZO_WIN_BLACKMAN: a[0] = 0.42; a[1] = 0.5; a[2] = 0.08; return 3;
    case SGZO_WIN_EXACT_BLACKMAN: a[0] = 7938.0 / 18608.0; a[1] = 9240.0 / 18608.0; a[2] = 1430.0 / 18608.0; return 3;
    case SGZO_WIN_NUTTALL: a[0] = 0.355768; a[1] = 0.487396; a[2] = 0.144232; a[3] = 0.012604; return 4;
    case SGZO_WIN_BLACKMAN_NUTTALL: a[0] = 0.3635819; a[1] = 0.4891775; a[2] = 0.1365995; a[3] = 0.0106411; return 4;
    case SGZO_WIN_BLACKMAN_HARRIS: a[0] = 0.35875; a[1] = 0.48829; a[2] = 0.14128; a[3] = 0.01168; return 4;
    case SGZO_WIN_FLATTOP: a[0] = 0.21557895; a[1] = 0.41663158; a[2] = 0.277263158; a[3] = 0.083578947; a[4] = 0.006947368; return 5;
    default: a[0] = 1.0; return 1;
    }
}

/* CComplexResonator::Constant::mapSystemHz(mappedHz, size, numVectors, sampleRate, qIsFree, vectorLength = 8, windowSize)
 * (call site: TransformConstant::remapResonator, Source/Spectrum/TransformConstant.h:120-123).  UNVERIFIED vs cpl.
 *   filter i is centred on mappedHz[i]; its resolution is the spacing to its neighbour (the last filter reuses the spacing before it);
 *   the spacing in Hz is the bandwidth B_i; "Q not free" bounds the equivalent window length fs / B_i by the window size
 *   (SpectrumParameters.h: freeQ "frees the quality factor from being bounded by the window size");
 *   vectors m = -(K-1) .. K-1 are detuned by m B_i;  gain_i = 1 - r_i  (a full-scale sine on the centre reads 1/2 unwindowed,
 *   the two-sided DFT convention of the FFT branch).
 * layout: coeff[(v * P + i)] for v = m + K - 1;  everything computed in double and rounded once. */
void sgzo_resonator_map(const sgzo_spectrum_params *p, const float *mapped, sgzo_cf *coeff, float *gain, float *weights, int *vectors)
{
    const uint32_t P = p->axis_points;
    double a[SGZO_RES_MAX_TERMS];
    const int K = sgzo_window_cosine_terms(p->window_type, a);
    const int V = 2 * K - 1;
    const double fs = (double)p->sample_rate;
    *vectors = V;
    for (int v = 0; v < V; ++v) {
        const int m = v - (K - 1), am = m < 0 ? -m : m;
        const double w = am == 0 ? a[0] : ((am & 1) ? -0.5 : 0.5) * a[am];
        weights[v] = (float)w;
    }
    for (uint32_t i = 0; i < P; ++i) {
        const uint32_t k = i + 1 >= P ? P - 2 : i;
        double hDiff = fabs((double)mapped[k + 1] - (double)mapped[k]);
        double length = hDiff > 0 ? fs / hDiff : (double)p->window_size;           /* equivalent window length in samples */
        if (!p->free_q && length > (double)p->window_size) length = (double)p->window_size;
        if (length < 2.0) length = 2.0;
        const double B = fs / length;                                              /* -3 dB bandwidth, Hz */
        const double r = exp(-M_PI * B / fs);
        gain[i] = (float)(1.0 - r);
        for (int v = 0; v < V; ++v) {
            const double omega = 2.0 * M_PI * ((double)mapped[i] + (double)(v - (K - 1)) * B) / fs;
            coeff[(size_t)v * P + i].re = (float)(r * cos(omega));
            coeff[(size_t)v * P + i].im = (float)(r * sin(omega));
        }
    }
}

/* CComplexResonator::resonateReal<V>(constant, data, channels, numSamples) (call sites TransformDSP.inl:1254-1291): every sample
 * advances every resonator of every vector, fp32 (T = float).  UNVERIFIED vs cpl: the operation order of the complex product.
 * state layout: [(signal * V + v) * P + i] */
void sgzo_resonate_real(const sgzo_cf *coeff, uint32_t P, int V, sgzo_cf *state, const float *const *work, int signals, size_t n)
{
    for (int s = 0; s < signals; ++s)
        for (int v = 0; v < V; ++v)
            for (uint32_t i = 0; i < P; ++i) {
                const float cr = coeff[(size_t)v * P + i].re, ci = coeff[(size_t)v * P + i].im;
                float re = state[((size_t)s * V + v) * P + i].re, im = state[((size_t)s * V + v) * P + i].im;
                for (size_t t = 0; t < n; ++t) {
                    const float nre = (re * cr - im * ci) + work[s][t];
                    const float nim = re * ci + im * cr;
                    re = nre; im = nim;
                }
                state[((size_t)s * V + v) * P + i].re = re;
                state[((size_t)s * V + v) * P + i].im = im;
            }
}

/* TransformPair::resonatingDispatch, Source/Spectrum/TransformDSP.inl:1213-1295: which signals the resonators see.  Note that Mid is
 * left + right and MidSide is (left - right, left + right) -- no halving and side first, unlike the FFT branch (prepareTransform). */
int sgzo_resonator_dispatch(uint32_t mode, const float *L, const float *R, size_t n, float *work0, float *work1)
{
    switch (mode) {
    case SGZO_CH_RIGHT: for (size_t i = 0; i < n; ++i) work0[i] = R[i]; return 1;                       /* :1250-1256 */
    case SGZO_CH_LEFT: for (size_t i = 0; i < n; ++i) work0[i] = L[i]; return 1;                        /* :1257-1262 */
    case SGZO_CH_MERGE: for (size_t i = 0; i < n; ++i) work0[i] = L[i] + R[i]; return 1;                /* :1263-1268 */
    case SGZO_CH_SIDE: for (size_t i = 0; i < n; ++i) work0[i] = L[i] - R[i]; return 1;                 /* :1269-1274 */
    case SGZO_CH_MIDSIDE:                                                                              /* :1275-1280 */
        for (size_t i = 0; i < n; ++i) { work0[i] = L[i] - R[i]; work1[i] = L[i] + R[i]; }
        return 2;
    default:                                                                                           /* Phase, Separate, Complex :1281-1293 */
        for (size_t i = 0; i < n; ++i) { work0[i] = L[i]; work1[i] = R[i]; }
        return 2;
    }
}

/* CComplexResonator::getWholeWindowedState<ISA>(constant, windowType, out, outChannels, numFilters) (TransformDSP.inl:1152-1161),
 * then the RSNT branch of mapToLinearSpace (:1103-1133).  out = csp [2P] complex: signal 0 at [0, P), signal 1 at [P, 2P)
 * (what mapAndTransformDFTFilters reads: newVals[i*2] and newVals[i*2 + size*2]).  UNVERIFIED vs cpl: summation order (centre first,
 * then -m, +m outwards) and the gain.  Phase (:1111-1127): cpl::Math::square of a complex is taken as |z|^2, so that
 * cancellation = |L + R| (the FFT branch's definition, :825-830) -- the expression does not type-check otherwise. */
void sgzo_resonator_windowed_state(const sgzo_spectrum_params *p, const sgzo_cf *state, const float *gain, const float *weights,
                                   int V, int signals, sgzo_cf *csp)
{
    const uint32_t P = p->axis_points;
    const int K = (V + 1) / 2;
    for (int s = 0; s < signals; ++s)
        for (uint32_t i = 0; i < P; ++i) {
            const sgzo_cf *st = state + (size_t)s * V * P + i;
            float re = weights[K - 1] * st[(size_t)(K - 1) * P].re, im = weights[K - 1] * st[(size_t)(K - 1) * P].im;
            for (int m = 1; m < K; ++m) {
                re = re + weights[K - 1 - m] * st[(size_t)(K - 1 - m) * P].re;
                im = im + weights[K - 1 - m] * st[(size_t)(K - 1 - m) * P].im;
                re = re + weights[K - 1 + m] * st[(size_t)(K - 1 + m) * P].re;
                im = im + weights[K - 1 + m] * st[(size_t)(K - 1 + m) * P].im;
            }
            csp[(size_t)s * P + i].re = re * gain[i];
            csp[(size_t)s * P + i].im = im * gain[i];
        }
    if (p->channel_mode == SGZO_CH_PHASE) {
        for (uint32_t x = 0; x < P; ++x) {
            const sgzo_cf l = csp[x], r = csp[x + P];
            const float sr = l.re + r.re, si = l.im + r.im;
            const float cancellation = sqrtf(sr * sr + si * si);
            const float mid = sqrtf(l.re * l.re + l.im * l.im) + sqrtf(r.re * r.re + r.im * r.im);
            csp[x].re = mid;
            csp[x].im = 1.0f - (mid > 0 ? cancellation / mid : 0.0f);
        }
    }
}

/* frames of an offline RSNT render: one per sampleBufferSize samples (audioEntryPoint :1172-1201) */
long sgzo_resonator_num_frames(size_t nsamples, uint32_t hop) { return hop ? (long)(nsamples / hop) : 0; }

/* Offline job in RSNT mode: audioEntryPoint (:1165-1211) fed the whole buffer, resonators starting from rest
 * (TransformPair.h:183 resetState), one frame per `hop` samples, then addAudioFrame -> mapAndTransformDFTFilters -> blend. */
long sgzo_resonator_spectrogram(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                                uint8_t *rgba_out, sgzo_cf *line_out, sgzo_cf *mapped_out)
{
    return sgzo_resonator_spectrogram_scaled(p, planar, nsamples, rgba_out, line_out, mapped_out, NULL);
}

/* ... and, for the tests' error bars, the size of what the window kernel sums: scale_out [F][C][2][P] = gain sum_v |w_v| |s_v| per signal.
 * The windowed value is a difference of nearly equal terms when the bandwidth is small against the detuning (long resonators, free
 * Q): two correct fp32 evaluations of the recurrence differ by a few eps of THIS quantity, not of the result. */
long sgzo_resonator_spectrogram_scaled(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                                       uint8_t *rgba_out, sgzo_cf *line_out, sgzo_cf *mapped_out, float *scale_out)
{
    const uint32_t P = p->axis_points, C = p->num_pairs, hop = p->hop;
    const long F = sgzo_resonator_num_frames(nsamples, hop);
    if (F <= 0 || P < 2) return 0;
    float *mapped = (float *)malloc(sizeof(float) * P), *slope = (float *)malloc(sizeof(float) * P);
    float *gain = (float *)malloc(sizeof(float) * P), weights[2 * SGZO_RES_MAX_TERMS];
    float ratios[SGZO_NUM_SPEC_COLOURS + 1];
    sgzo_cf *coeff = (sgzo_cf *)malloc(sizeof(sgzo_cf) * P * (2 * SGZO_RES_MAX_TERMS - 1));
    int V = 1;
    sgzo_remap_frequencies(p, mapped);
    sgzo_slope_map(p, mapped, slope);
    sgzo_colour_ratios(p->ratios, ratios);
    sgzo_resonator_map(p, mapped, coeff, gain, weights, &V);
    sgzo_cf *state = (sgzo_cf *)calloc((size_t)C * 2 * V * P, sizeof(sgzo_cf));
    sgzo_cf *csp = (sgzo_cf *)calloc((size_t)P * 2, sizeof(sgzo_cf));
    sgzo_cf *states = (sgzo_cf *)calloc((size_t)C * SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
    sgzo_cf *results = (sgzo_cf *)calloc((size_t)C * SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
    sgzo_cf *frames = (sgzo_cf *)calloc((size_t)C * P, sizeof(sgzo_cf));
    float *work0 = (float *)malloc(sizeof(float) * hop), *work1 = (float *)malloc(sizeof(float) * hop);
    const int sc = p->channel_mode > SGZO_CH_SIDE ? 2 : 1;
    for (long f = 0; f < F; ++f) {
        for (uint32_t pr = 0; pr < C; ++pr) {
            const float *L = planar[2 * pr] + (size_t)f * hop, *R = planar[2 * pr + 1] + (size_t)f * hop;
            const float *work[2] = {work0, work1};
            const int signals = sgzo_resonator_dispatch(p->channel_mode, L, R, hop, work0, work1);
            sgzo_cf *st = state + (size_t)pr * 2 * V * P;
            sgzo_resonate_real(coeff, P, V, st, work, signals, hop);
            memset(csp, 0, sizeof(sgzo_cf) * (size_t)P * 2);
            sgzo_resonator_windowed_state(p, st, gain, weights, V, signals, csp);
            if (mapped_out) memcpy(mapped_out + ((size_t)f * C + pr) * 2 * P, csp, sizeof(sgzo_cf) * (size_t)P * sc);
            if (scale_out)
                for (int sg = 0; sg < 2; ++sg)
                    for (uint32_t i = 0; i < P; ++i) {
                        double acc = 0;
                        if (sg < signals)
                            for (int v = 0; v < V; ++v) {
                                const sgzo_cf z = st[((size_t)sg * V + v) * P + i];
                                acc += fabs((double)weights[v]) * hypot((double)z.re, (double)z.im);
                            }
                        scale_out[(((size_t)f * C + pr) * 2 + sg) * P + i] = (float)(acc * (double)gain[i]);
                    }
            sgzo_cf *fs = states + (size_t)pr * SGZO_NUM_GRAPHS * P, *rs = results + (size_t)pr * SGZO_NUM_GRAPHS * P;
            sgzo_map_and_transform_filters(p, slope, csp, fs, rs);
            memcpy(frames + (size_t)pr * P, rs, sizeof(sgzo_cf) * P);
            if (line_out) memcpy(line_out + ((size_t)f * C + pr) * SGZO_NUM_GRAPHS * P, rs, sizeof(sgzo_cf) * SGZO_NUM_GRAPHS * P);
        }
        if (rgba_out) sgzo_blend_column(p, ratios, frames, C, rgba_out + (size_t)f * P * 4);
    }
    free(mapped); free(slope); free(gain); free(coeff); free(state); free(csp); free(states); free(results); free(frames);
    free(work0); free(work1);
    return F;
}
