/*
 * scope_spectral.c -- CPU restatement of the Oscilloscope's spectral trigger and of its per-sample frequency colouring
 * (SURVEY 8(f) #3).  TEST INFRASTRUCTURE (see sgz_oracle.h).  Follows:
 *   Source/Oscilloscope/OscilloscopeDSP.inl:62-223    Oscilloscope::calculateFundamentalPeriod (Spectral branch, PHASE_VOCODER off)
 *   Source/Oscilloscope/OscilloscopeDSP.inl:231-308   Oscilloscope::calculateTriggeringOffset
 *   Source/Oscilloscope/OscilloscopeDSP.inl:445-517, :588-647  StreamState::audioProcessing: crossover -> band energies -> RGB
 *   Source/Oscilloscope/ChannelData.h:163-171         tuneCrossOver(300, 3000, sampleRate), tuneColourSmoothing(ms, sampleRate)
 *
 * What is NOT in the mounted tree and is restated from published definitions (UNVERIFIED vs cpl / DustFFT):
 *   signaldust::DustFFT_fwdDa        an unnormalised forward complex DFT on doubles (only |X| and a conjugation-invariant ratio
 *                                    of bins are used, so the sign convention does not matter);
 *   cpl::dsp::goertzel(buf, N, w)    the textbook Goertzel filter: s[n] = x[n] + 2 cos(w) s[n-1] - s[n-2] over the N samples,
 *                                    z = s[N-1] - exp(-i w) s[N-2] = exp(i w (N-1)) sum_n x[n] exp(-i w n), i.e. the phase is
 *                                    referenced to the LAST sample.  Evidence for this form: it is the one under which the
 *                                    reference's own "correct phase by delta" term (phase += record.offset * tau, :287) cancels
 *                                    exp(i w (N-1)) = exp(i (tau offset - w)) and the display starts exactly on the sine's rising
 *                                    zero crossing (tests/test_oracle_spectral.py); with a first-sample phase reference the same
 *                                    formula would leave the display tau * offset off;
 *   cpl::dsp::LinkwitzRileyNetwork<float, 3>   a 3-band Linkwitz-Riley (4th order) tree: band 0 = LP4_f1(x),
 *                                    rest = HP4_f1(x), band 1 = LP4_f2(rest), band 2 = HP4_f2(rest); LP4 / HP4 = two cascaded
 *                                    2nd-order Butterworth sections (Q = 1/sqrt 2, bilinear transform, RBJ form), each run as a
 *                                    transposed direct form II in fp32;
 *   cpl::dsp::SmoothedParameterState<float, 1>::design(ms, sr)   pole = exp(-1 / (ms / 1000 * sr));
 *   UPixel::lerp(other, t)           per component  (uint8)(a + (b - a) * t)  in the type of t.
 * std::nth_element is restated as libstdc++'s introselect (the reference's Linux build), because which of several records with the
 * same bin index lands in the middle decides the median record's fractional offset.
 */
#include "sgz_oracle.h"
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ spectral trigger */

enum { LOOKAHEAD = 8192, MEDIAN = 8 };

static double rec_omega(const sgzo_bin_record *r) { return (double)r->index + r->offset; }     /* BinRecord::omega(), Oscilloscope.h */

/* libstdc++ bits/stl_algo.h: std::nth_element(first, first + 4, last, by index) on 8 records */
static int rec_less(const sgzo_bin_record *a, const sgzo_bin_record *b) { return a->index < b->index; }
static void rec_swap(sgzo_bin_record *a, sgzo_bin_record *b) { sgzo_bin_record t = *a; *a = *b; *b = t; }
static void move_median_to_first(sgzo_bin_record *result, sgzo_bin_record *a, sgzo_bin_record *b, sgzo_bin_record *c)
{
    if (rec_less(a, b)) {
        if (rec_less(b, c)) rec_swap(result, b);
        else if (rec_less(a, c)) rec_swap(result, c);
        else rec_swap(result, a);
    } else if (rec_less(a, c)) rec_swap(result, a);
    else if (rec_less(b, c)) rec_swap(result, c);
    else rec_swap(result, b);
}
static sgzo_bin_record *unguarded_partition(sgzo_bin_record *first, sgzo_bin_record *last, sgzo_bin_record *pivot)
{
    for (;;) {
        while (rec_less(first, pivot)) ++first;
        --last;
        while (rec_less(pivot, last)) --last;
        if (!(first < last)) return first;
        rec_swap(first, last);
        ++first;
    }
}
static void insertion_sort(sgzo_bin_record *first, sgzo_bin_record *last)
{
    if (first == last) return;
    for (sgzo_bin_record *i = first + 1; i != last; ++i) {
        if (rec_less(i, first)) {
            sgzo_bin_record val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(*first));
            *first = val;
        } else {                                           /* __unguarded_linear_insert */
            sgzo_bin_record val = *i;
            sgzo_bin_record *next = i - 1, *cur = i;
            while (rec_less(&val, next)) { *cur = *next; cur = next; --next; }
            *cur = val;
        }
    }
}
void sgzo_nth_element_by_index(sgzo_bin_record *v, int n, int nth)
{
    sgzo_bin_record *first = v, *last = v + n, *nthp = v + nth;
    if (first == last || nthp == last) return;
    int depth = 0;
    for (int k = n; k > 1; k >>= 1) ++depth;               /* std::__lg(n) */
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            /* libstdc++ falls back to __heap_select here.  Unreachable for the reference's n = 8 (depth limit 2 lg 8 = 6, every
             * partition shortens the range, a range of <= 3 leaves the loop); a plain selection keeps the function total. */
            for (sgzo_bin_record *i = first; i <= nthp; ++i)
                for (sgzo_bin_record *j = i + 1; j < last; ++j)
                    if (rec_less(j, i)) rec_swap(i, j);
            return;
        }
        --depth;
        sgzo_bin_record *mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        sgzo_bin_record *cut = unguarded_partition(first + 1, last, first);
        if (cut <= nthp) first = cut;
        else last = cut;
    }
    insertion_sort(first, last);
}

static double eval_at(const float *a, const float *b, int mode, long p)
{
    if (mode == 1) return (double)(0.5f * (a[p] + b[p]));
    if (mode == 2) return (double)(0.5f * (a[p] - b[p]));
    return (double)a[p];
}

/* complex helpers on sgzo_cd */
static sgzo_cd c_sub(sgzo_cd a, sgzo_cd b) { sgzo_cd r = {a.re - b.re, a.im - b.im}; return r; }

/* quadDelta, OscilloscopeDSP.inl:113-124: real((xm1 - x1) / (2 x0 - xm1 - x1)); std::complex<double> division restated as the
 * textbook (a + ib) / (c + id) = ((ac + bd) + i(bc - ad)) / (c^2 + d^2) (libstdc++ without -ffast-math calls __divdc3, which agrees
 * with this except for its overflow / NaN recovery) */
static double quad_delta(const sgzo_cd *X, size_t w)
{
    const sgzo_cd x0 = X[w], x1 = X[w + 1], xm1 = X[w == 0 ? 1 : w - 1];
    sgzo_cd denom = {x0.re * 2.0 - xm1.re - x1.re, x0.im * 2.0 - xm1.im - x1.im};
    if ((denom.re + denom.im) == 0) return 0;
    const sgzo_cd num = c_sub(xm1, x1);
    return (num.re * denom.re + num.im * denom.im) / (denom.re * denom.re + denom.im * denom.im);
}

/* calculateFundamentalPeriod, Spectral branch.  mem: ring memory of `size` samples with write cursor `cursor` (eval.startFrom(-k) reads
 * from cursor - k, wrapping at size). */
void sgzo_scope_fundamental(sgzo_spectral_state *ts, const float *memA, const float *memB, int eval_mode, size_t size, size_t cursor,
                            double window_size, double sample_rate, double threshold, double hysteresis)
{
    static sgzo_cd transformBuffer[LOOKAHEAD];
    const size_t TransformSize = LOOKAHEAD;
    if (size == 0) return;                                                                  /* !eval.isWellDefined() */
    size_t offset = (size_t)ceil(window_size);
    if (offset < LOOKAHEAD) offset = LOOKAHEAD;                                             /* :92 */
    long p = ((long)cursor - (long)offset) % (long)size;
    if (p < 0) p += (long)size;
    for (size_t i = 0; i < LOOKAHEAD; ++i) {
        transformBuffer[i].re = eval_at(memA, memB, eval_mode, p);
        transformBuffer[i].im = 0;
        if (++p == (long)size) p = 0;
    }
    sgzo_fft_forward_f64(transformBuffer, (uint32_t)TransformSize);
    const double quarterSemitone = pow(2, 0.25 / 12.0) - 1;
    const double invHysteresis = 1 - hysteresis;
    sgzo_bin_record max;
    max.index = 1;
    max.value = fmax(threshold * (double)TransformSize / 6.0, hypot(transformBuffer[1].re, transformBuffer[1].im));
    max.offset = quad_delta(transformBuffer, 1);
    for (size_t i = 2; i < (TransformSize >> 1); ++i) {
        sgzo_bin_record current = {i, hypot(transformBuffer[i].re, transformBuffer[i].im), 0};
        if (invHysteresis * current.value > max.value * 2) {                                /* candidate must be vastly better */
            if (rec_omega(&max) > 0) {
                current.offset = quad_delta(transformBuffer, i);
                const double factor = rec_omega(&current) / rec_omega(&max);
                const double sensivity = current.value / max.value;
                if (invHysteresis * sensivity > 20) { max = current; continue; }
                if (fabs(1 - factor) < invHysteresis * quarterSemitone) { max = current; continue; }
                const double multipleDeviation = fabs(factor - floor(factor + 0.5));
                if (invHysteresis * fabs(multipleDeviation) > quarterSemitone) max = current;
            } else {
                max = current;
                max.offset = quad_delta(transformBuffer, max.index);
            }
        }
    }
    sgzo_bin_record localMedian[MEDIAN];
    memcpy(localMedian, ts->median, sizeof(localMedian));                                   /* copy old filter */
    ts->median[ts->median_pos] = max;                                                       /* store new data */
    ts->median_pos = (ts->median_pos + 1) & (MEDIAN - 1);
    sgzo_nth_element_by_index(localMedian, MEDIAN, MEDIAN >> 1);
    const sgzo_bin_record *oldMedianBin = &localMedian[MEDIAN >> 1];
    if (oldMedianBin->index != (uint64_t)-1 && fabs(rec_omega(&max) - rec_omega(oldMedianBin)) > 0.5) max = *oldMedianBin;
    ts->record = max;
    double fundamental = sample_rate * rec_omega(&max) / (double)TransformSize;
    ts->fundamental = fundamental = fmax(5.0, fundamental);
    ts->cycle_samples = sample_rate / fundamental;
}

/* calculateFundamentalPeriod with state.customTrigger (OscilloscopeDSP.inl:71-81): the user names the frequency; no transform, the
 * median filter is not touched.  sgzo_scope_trigger_offset follows as usual. */
void sgzo_scope_fundamental_custom(sgzo_spectral_state *ts, double custom_frequency, double sample_rate)
{
    const double normalizedFrequency = custom_frequency / sample_rate;
    ts->record.index = 0; ts->record.value = 1; ts->record.offset = normalizedFrequency * (double)LOOKAHEAD;
    ts->fundamental = custom_frequency;
    ts->cycle_samples = sample_rate / custom_frequency;
}

/* calculateTriggeringOffset, Spectral branch */
void sgzo_scope_trigger_offset(sgzo_spectral_state *ts, const float *memA, const float *memB, int eval_mode, size_t size, size_t cursor,
                               double window_size, double sample_rate, double phase_offset_degrees)
{
    const size_t TransformSize = LOOKAHEAD;
    if (size == 0) return;
    const double tau = 6.283185307179586476925286766559;
    const double radians = tau * rec_omega(&ts->record) / (double)TransformSize;
    const double offsetReal = fmax((double)LOOKAHEAD, window_size + ts->cycle_samples);
    const size_t offset = (size_t)ceil(offsetReal);
    const double sampleDifference = (double)offset - (window_size + ts->cycle_samples);
    long p = ((long)cursor - (long)offset) % (long)size;
    if (p < 0) p += (long)size;
    /* cpl::dsp::goertzel(temporaryBuffer, LookaheadSize, radians) (UNVERIFIED, see the header) */
    const double coeff = 2 * cos(radians);
    double s1 = 0, s2 = 0;
    for (size_t n = 0; n < LOOKAHEAD; ++n) {
        const double x = eval_at(memA, memB, eval_mode, p);
        const double s0 = x + coeff * s1 - s2;
        s2 = s1; s1 = s0;
        if (++p == (long)size) p = 0;
    }
    const double zr = s1 - s2 * cos(radians), zi = s2 * sin(radians);
    const double rotation = -sampleDifference * radians;
    const double cr = cos(rotation), ci = -sin(rotation);                                   /* z *= complex(cos(rotation), -sin(rotation)) */
    const double wr = zr * cr - zi * ci, wi = zr * ci + zi * cr;
    double phase = tau - atan2(wi, wr);
    phase += ts->record.offset * tau;
    phase -= 1.5707963267948966192313216916398;
    phase += tau * phase_offset_degrees / 360;
    phase = fmod(phase, tau);
    while (phase < 0) phase += tau;
    ts->phase = phase;
    const double cycles = phase / tau;
    ts->sample_offset = cycles * sample_rate / ts->fundamental - 1;
}

/* ------------------------------------------------------------------------------------------------ frequency colouring */

static void butterworth(double fnorm, int highpass, float c[5])
{
    const double w0 = 6.283185307179586476925286766559 * fnorm;
    const double cw = cos(w0), alpha = sin(w0) / (2.0 * 0.70710678118654752440);
    const double a0 = 1 + alpha;
    double b0, b1;
    if (highpass) { b0 = (1 + cw) / 2; b1 = -(1 + cw); }
    else { b0 = (1 - cw) / 2; b1 = 1 - cw; }
    c[0] = (float)(b0 / a0); c[1] = (float)(b1 / a0); c[2] = (float)(b0 / a0);
    c[3] = (float)(-2 * cw / a0); c[4] = (float)((1 - alpha) / a0);
}

/* Crossover::Coefficients::design({low / sampleRate, high / sampleRate}) with float arguments (ChannelData.h:165) */
void sgzo_lr_design(double low_hz, double high_hz, double sample_rate, sgzo_lr_coeffs *out)
{
    const float f1 = (float)(low_hz / sample_rate), f2 = (float)(high_hz / sample_rate);
    butterworth((double)f1, 0, out->lp1); butterworth((double)f1, 1, out->hp1);
    butterworth((double)f2, 0, out->lp2); butterworth((double)f2, 1, out->hp2);
}

static float biquad(const float c[5], float z[2], float x)
{
    const float y = c[0] * x + z[0];
    z[0] = (c[1] * x - c[3] * y) + z[1];
    z[1] = c[2] * x - c[4] * y;
    return y;
}

/* network.process(x, coeffs) -> BandArray */
void sgzo_lr_process(sgzo_lr_state *st, const sgzo_lr_coeffs *k, float x, float bands[3])
{
    const float low = biquad(k->lp1, st->z[1], biquad(k->lp1, st->z[0], x));
    const float rest = biquad(k->hp1, st->z[3], biquad(k->hp1, st->z[2], x));
    bands[0] = low;
    bands[1] = biquad(k->lp2, st->z[5], biquad(k->lp2, st->z[4], rest));
    bands[2] = biquad(k->hp2, st->z[7], biquad(k->hp2, st->z[6], rest));
}

float sgzo_colour_smooth_pole(double milliseconds, double sample_rate) { return (float)exp(-1.0 / (milliseconds / 1000.0 * sample_rate)); }

/* filterStates, OscilloscopeDSP.inl:460-468 */
void sgzo_colour_filter_states(const float bands[3], float states[3], float pole)
{
    for (int i = 0; i < 3; ++i) {
        const float input = bands[i] * bands[i];
        states[i] = input + pole * (states[i] - input);
    }
}

static uint8_t to_u8(float v)
{
    /* static_cast<uint8_t>(float): truncation; out-of-range and NaN (all-zero state: 255 / 0 * 0) are undefined in C++ -- x86's
     * cvttss2si yields 0x80000000, whose low byte is 0, which is what this restatement returns */
    if (!(v > -1.0f) || !(v < 256.0f)) return 0;
    return (uint8_t)v;
}
static uint8_t lerp_u8f(uint8_t a, uint8_t b, float t) { return to_u8((float)a + ((float)b - (float)a) * t); }

/* accumulateColour, OscilloscopeDSP.inl:472-497.  colours[band][rgb], key / out: RGBA8 */
void sgzo_colour_accumulate(const float state[3], const float colours[3][3], const uint8_t key[4], float blend, uint8_t out[4])
{
    const float PixelMax = 255.0f;
    float red = 0, green = 0, blue = 0;
    for (int i = 0; i < 3; ++i) {
        red += state[i] * colours[i][0];
        green += state[i] * colours[i][1];
        blue += state[i] * colours[i][2];
    }
    const float invMax = PixelMax / fmaxf(red, fmaxf(blue, green));
    const uint8_t ret[4] = {to_u8(red * invMax), to_u8(green * invMax), to_u8(blue * invMax), 255};
    for (int k = 0; k < 4; ++k) out[k] = lerp_u8f(ret[k], key[k], blend);
}

/* currentColour.lerp(nextColour, delta) with a double delta (OscilloscopeRendering.cpp:876) */
void sgzo_colour_lerp_f64(const uint8_t a[4], const uint8_t b[4], double t, uint8_t out[4])
{
    for (int k = 0; k < 4; ++k) {
        const double v = (double)a[k] + ((double)b[k] - (double)a[k]) * t;
        out[k] = (!(v > -1.0) || !(v < 256.0)) ? 0 : (uint8_t)v;
    }
}
