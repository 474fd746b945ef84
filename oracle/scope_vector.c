/*
 * scope_vector.c -- CPU restatement of the Oscilloscope and Vectorscope arithmetic on the hot path.
 * TEST INFRASTRUCTURE (see sgz_oracle.h).  Follows:
 *   Source/Oscilloscope/StreamPreprocessing.h:315-349   ZeroCrossingProcessor::process
 *   Source/Oscilloscope/OscilloscopeDSP.inl:311-385     executeSamplingWindows (trigger channel mix)
 *   Source/Oscilloscope/OscilloscopeDSP.inl:231-240     calculateTriggeringOffset (ZeroCrossing)
 *   Source/Oscilloscope/OscilloscopeRendering.cpp:551-649,:790-891  drawWavePlot, Lanczos branch
 *   Source/Oscilloscope/OscilloscopeDSP.inl:713-886 / Source/Vectorscope/VectorscopeRendering.cpp:826-889 runPeakFilter
 *   Source/Vectorscope/VectorscopeRendering.cpp:500-746 drawPolarPlot
 *   Source/Vectorscope/Vectorscope.cpp:268-377          Processor::audioProcessing
 * cpl::simd::{atan,sincos,cos} are restated with libm (UNVERIFIED vs cpl: cpl uses SIMD polynomial
 * approximations; parity on these outputs is a stated fp32 tolerance, never bit-exactness).
 */
#include "sgz_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* OscChannels, Source/Common/CommonSignalizer.h:458-493 */
enum { OSC_LEFT = 0, OSC_RIGHT = 1, OSC_MID = 2, OSC_SIDE = 3, OSC_SEPARATE = 4, OSC_MIDSIDE = 5 };

/* ZeroCrossingProcessor::process over one block of the trigger channel.
 * a,b = the trigger pair's two channels (b unused for Left/Right/Separate: pass the chosen channel as a). */
size_t sgzo_zero_crossing_process(sgzo_zero_crossing_state *st, uint32_t osc_mode, const float *a,
                                  const float *b, size_t n, uint64_t *out, size_t max_out)
{
    size_t produced = 0;
    for (size_t i = 0; i < n; ++i) {
        double sample;
        switch (osc_mode) {
        case OSC_MID: case OSC_MIDSIDE: sample = (double)(0.5f * (a[i] + b[i])); break;   /* :371-376, MidSide -> Mid :342-352 */
        case OSC_SIDE: sample = (double)(0.5f * (a[i] - b[i])); break;                    /* :377-382 */
        default: sample = (double)a[i]; break;
        }
        if (sample > 0 && st->state < 0) {                 /* StreamPreprocessing.h:333-337 */
            st->armed = 1;
            st->cross_origin = st->steady_clock + st->count;
        }
        if (st->armed && sample > st->threshold) {         /* :339-343 */
            st->armed = 0;
            if (produced < max_out) out[produced] = st->cross_origin;
            produced++;
        }
        st->state = sample;
        st->count++;
    }
    return produced;
}

/* Scalars of drawWavePlot for TriggeringMode::ZeroCrossing (OscilloscopeRendering.cpp:551-600, :790-826) */
typedef struct { double samplePos0, inc, samplesPerPixel, unit0, right; } scope_scalars;

static scope_scalars scope_derive(const sgzo_scope_view *v)
{
    scope_scalars s;
    const double horizontalDelta = v->right - v->left;
    const double sizeMinusOne = fmax(1.0, v->window_size - 1);
    const double pixelsPerSample = v->rendering_scale * fabs(((double)v->width - 1) / (sizeMinusOne * horizontalDelta));
    /* calculateTriggeringOffset, OscilloscopeDSP.inl:238 */
    const double sampleOffset = (v->window_size * 0.5 - (double)(int)(v->window_size * 0.5)) - 1.5;
    s.inc = horizontalDelta / (v->rendering_scale * ((double)v->width - 1));
    s.samplesPerPixel = 1.0 / pixelsPerSample;
    s.unit0 = v->left;
    s.right = v->right;
    s.samplePos0 = sampleOffset + (-s.unit0 / s.inc * s.samplesPerPixel);   /* :802-826 */
    return s;
}

size_t sgzo_scope_num_points(const sgzo_scope_view *v)
{
    const scope_scalars s = scope_derive(v);
    size_t n = 0;
    double unitSpacePos = s.unit0;
    do { unitSpacePos += s.inc; ++n; } while (unitSpacePos < (s.right + s.inc));
    return n;
}

/* Lanczos branch of drawWavePlot, OscilloscopeRendering.cpp:790-891.
 * `ring` is the channel's front buffer in time order with ring[0] at the stream cursor (the oldest
 * sample; CLIFOStream proxy view, begin()+cursorPosition()).  DynamicChannelEvaluator::startFrom
 * (SampleColourEvaluators.h:75-95) positions the read pointer at cursor + offset and wraps it
 * circularly over the view, inc() wraps likewise; offset = -floor(samplePos) - KernelSize (:829).
 * UNVERIFIED vs cpl: CLIFOStream::createProxyView()/cursorPosition() semantics. */
size_t sgzo_scope_lanczos(const sgzo_scope_view *v, const float *ring, size_t len,
                          float *out_x, float *out_y, size_t max_points)
{
    enum { KernelSize = 10, KernelBufferSize = 21 };
    if (len == 0) return 0;
    const scope_scalars s = scope_derive(v);
    double samplePos = s.samplePos0;
    double currentSample = floor(samplePos);
    double unitSpacePos = s.unit0;
    long cursor = (-(long)floor(samplePos) - KernelSize) % (long)len;
    if (cursor < 0) cursor += (long)len;
    float kernel[KernelBufferSize];
    for (int i = 0; i < KernelBufferSize; ++i) {
        kernel[i] = ring[cursor];
        if (++cursor == (long)len) cursor = 0;
    }
    size_t n = 0;
    do {
        double delta = currentSample - samplePos;
        while (delta > 1) {
            samplePos += 1;
            delta -= 1;
            memmove(kernel, kernel + 1, sizeof(float) * (KernelBufferSize - 1));     /* std::rotate + overwrite */
            kernel[KernelBufferSize - 1] = ring[cursor];
            if (++cursor == (long)len) cursor = 0;
        }
        const double y = sgzo_lanczos_filter_f64(kernel, KernelBufferSize, (double)KernelSize + delta, KernelSize);
        if (n < max_points) { out_x[n] = (float)unitSpacePos; out_y[n] = (float)y; }
        ++n;
        currentSample += s.samplesPerPixel;
        unitSpacePos += s.inc;
    } while (unitSpacePos < (s.right + s.inc));
    return n;
}

/* runPeakFilter: max|x| over the window with the SIMD tail dropped, then
 * env = max(env*coeff, peak^2), gain = 1/max_c sqrt(env_c).  `coeff_pow` is the already exponentiated
 * per-frame coefficient (VectorscopeRendering.cpp:838-842 / OscilloscopeDSP.inl:745-747). */
double sgzo_peak_filter(const float *const *ch, uint32_t nch, size_t n, uint32_t lanes,
                        double coeff_pow, double *env)
{
    const size_t stop = n - (n & (size_t)(lanes - 1));
    double start = 0;
    for (uint32_t c = 0; c < nch; ++c) {
        float peak = 0.0f;
        for (size_t i = 0; i < stop; ++i) { const float a = fabsf(ch[c][i]); if (a > peak) peak = a; }
        const double highest = (double)peak;
        /* filters.envelope[] is float storage (relaxed_atomic<AFloat>) */
        const float e = (float)fmax((double)(float)env[c] * coeff_pow, highest * highest);
        env[c] = (double)e;
        start = fmax(start, sqrt((double)e));
    }
    return 1.0 / start;
}

/* drawPolarPlot, one contiguous section, lanes = 8 (AVX fp32).  xyz = (X*len, Y*len, fade-1). */
void sgzo_vector_polar(const float *L, const float *R, size_t n, int fade, float *xyz)
{
    (void)fade;
    const float cosineRotation = -0.70710678118654752440f;   /* consts::sqrt_half_two_minus */
    const float sineRotation = 0.70710678118654752440f;      /* consts::sqrt_half_two */
    const long V = 8;
    const float fadePerSample = 1.0f / (float)n;
    const float incremental = fadePerSample * (float)V;
    float sampleFade[8], outFade[8];
    for (int l = 0; l < V; ++l) { sampleFade[l] = fadePerSample * (float)l; outFade[l] = fadePerSample * (float)l; }
    long i = 0;
    const long sectionSamples = (long)n;
    for (; i < sectionSamples - V; i += V) {
        for (int l = 0; l < V; ++l) {
            const float vl = L[i + l], vr = R[i + l];
            const float length = fmaxf(fabsf(vl), fabsf(vr));
            const float vY = vl * cosineRotation - vr * sineRotation;
            const float vX = vl * sineRotation + vr * cosineRotation;
            float angle = atanf(vX / vY);
            if (vl == 0.0f && vr == 0.0f) angle = 0.0f;
            const float sx = sinf(angle), cy = cosf(angle);
            outFade[l] = sampleFade[l] - 1.0f;
            xyz[(i + l) * 3 + 0] = sx * length;
            xyz[(i + l) * 3 + 1] = cy * length;
            xyz[(i + l) * 3 + 2] = outFade[l];
            sampleFade[l] += incremental;
        }
    }
    long remaining = 0;
    const float currentSampleFade = outFade[V - 1];
    for (; i < sectionSamples; ++i, ++remaining) {
        const float vl = L[i], vr = R[i];
        const float length = fmaxf(fabsf(vl), fabsf(vr));
        const float vY = vl * cosineRotation - vr * sineRotation;
        const float vX = vl * sineRotation + vr * cosineRotation;
        float angle = atanf(vX / vY);
        if (vl == 0.0f && vr == 0.0f) angle = 0.0f;
        xyz[i * 3 + 0] = sinf(angle) * length;
        xyz[i * 3 + 1] = cosf(angle) * length;
        xyz[i * 3 + 2] = currentSampleFade - (float)remaining * fadePerSample;
    }
}

/* VectorScope::Processor::audioProcessing, Vectorscope.cpp:268-377 (channels 0,1 only; tail dropped). */
void sgzo_vector_audio_processing(sgzo_vector_filters *f, const float *L, const float *R, size_t n,
                                  uint32_t lanes, float envelope_coeff, float stereo_coeff,
                                  float second_speed, int env_mode, float *gain_out)
{
    float filterEnv[2] = { f->env[0], f->env[1] };
    float balance[2][2] = { { f->balance[0][0], f->balance[0][1] }, { f->balance[1][0], f->balance[1][1] } };
    float phase[2] = { f->phase[0], f->phase[1] };
    const float stereoPoles[2] = { stereo_coeff, powf(stereo_coeff, second_speed) };
    const float envelope = envelope_coeff;
    const float mReal = -0.70710678118654752440f, mImag = 0.70710678118654752440f;
    n -= n & (size_t)(lanes - 1);
    for (size_t z = 0; z < n; ++z) {
        const float l = L[z], r = R[z];
        const float vX = l * mReal - r * mImag;
        const float vY = r * mImag + l * mReal;
        const float radians = atanf(vY / vX);
        const float angle = (vX == 0.0f && vY == 0.0f) ? 0.78539816339744830962f : radians;
        const float outPhase = cosf(angle * 2.0f);
        const float lSquared = l * l, rSquared = r * r;
        filterEnv[0] = lSquared + envelope * (filterEnv[0] - lSquared);
        filterEnv[1] = rSquared + envelope * (filterEnv[1] - rSquared);
        balance[0][0] = lSquared + stereoPoles[0] * (balance[0][0] - lSquared);
        balance[0][1] = rSquared + stereoPoles[0] * (balance[0][1] - rSquared);
        balance[1][0] = lSquared + stereoPoles[1] * (balance[1][0] - lSquared);
        balance[1][1] = rSquared + stereoPoles[1] * (balance[1][1] - rSquared);
        phase[0] = outPhase + stereoPoles[0] * (phase[0] - outPhase);
        phase[1] = outPhase + stereoPoles[1] * (phase[1] - outPhase);
    }
    if (env_mode == 1) {
        /* std::sqrt(T) with T = float (Vectorscope.cpp:351) */
        const double currentEnvelope = 1.0 / (double)fmaxf(sqrtf(filterEnv[0]), sqrtf(filterEnv[1]));
        f->env[0] = filterEnv[0]; f->env[1] = filterEnv[1];
        if (isnormal(currentEnvelope) && gain_out) *gain_out = (float)currentEnvelope;
    }
    for (int i = 0; i < 2; ++i) {
        f->phase[i] = phase[i];
        for (int j = 0; j < 2; ++j) f->balance[i][j] = balance[i][j];
    }
}

/* drawWavePlot (OscilloscopeRendering.cpp:551-891) for TriggeringMode None (0) / ZeroCrossing (4) and SubSampleInterpolation
 * Linear (2) / Lanczos (3), one evaluator (SampleColourEvaluators.h): eval_mode 0 = *A, 1 = 0.5 (A + B), 2 = 0.5 (A - B).
 * memA / memB: front-buffer memory (proxy view begin()) of `size` samples, `cursor` = cursorPosition().  xyz: (x, y, 0) per vertex
 * as handed to PrimitiveDrawer::addVertex.  Returns the vertex count. */
static float wave_eval(const float *a, const float *b, int mode, long idx)
{
    if (mode == 1) return 0.5f * (a[idx] + b[idx]);
    if (mode == 2) return 0.5f * (a[idx] - b[idx]);
    return a[idx];
}

/* Spectral triggering adds triggerState.cycleSamples / sampleOffset (:598-612, :808-811); colour_mem != NULL is
 * state.colourChannelsByFrequency: the evaluator's colour ring (colourData of the channel for Left / Right, auxColourData for Mid /
 * Side, SampleColourEvaluators.h:64,183) read at the audio position, RGBA8 per vertex into rgba (Linear :664-677: the sample's
 * colour; Lanczos :836-877: currentColour.lerp(nextColour, delta), the colours of the two newest kernel samples). */
size_t sgzo_scope_wave_plot_ex(const sgzo_scope_view *v, int trigger_mode, int interpolation, const float *memA, const float *memB,
                               int eval_mode, size_t size, size_t cursor, double cycle_samples, double sample_offset,
                               const uint32_t *colour_mem, float *xyz, uint32_t *rgba, size_t max_points)
{
    return sgzo_scope_wave_plot_ex2(v, trigger_mode, interpolation, memA, memB, eval_mode, size, cursor, cycle_samples, sample_offset, 0,
                                    0xffffffffu, colour_mem, xyz, rgba, max_points);
}

/* ... plus the remaining branches of drawWavePlot: TriggeringMode::Window (2) places the window by the host transport
 * (cs.transportPosition = playhead.getPositionInSamples() + numSamples, OscilloscopeDSP.inl:706; :588-592, :798-801), EnvelopeHold (3)
 * draws like ZeroCrossing (:593-597, :802-805); SubSampleInterpolation::None (0) draws the Linear vertex list as GL_POINTS
 * (dotSamples, :652-700, and no case in the switch), Rectangular (1) two vertices per sample with the previous and the current
 * sample's colour (:746-789; `key` = evaluator.getDefaultKey() when colours do not follow the frequency). */
size_t sgzo_scope_wave_plot_ex2(const sgzo_scope_view *v, int trigger_mode, int interpolation, const float *memA, const float *memB,
                                int eval_mode, size_t size, size_t cursor, double cycle_samples, double sample_offset,
                                int64_t transport_position, uint32_t key, const uint32_t *colour_mem, float *xyz, uint32_t *rgba,
                                size_t max_points)
{
    enum { KernelSize = 10, KernelBufferSize = 21 };
    if (size == 0) return 0;
    const double horizontalDelta = v->right - v->left;
    long roundedWindow = (long)ceil(v->window_size);                                   /* :563 */
    long quantizedCycleSamples = 0;
    const double sizeMinusOne = fmax(1.0, v->window_size - 1);
    const double pixelsPerSample = v->rendering_scale * fabs(((double)v->width - 1) / (sizeMinusOne * horizontalDelta));
    if (pixelsPerSample < 1 && interpolation != 0) interpolation = 2;                  /* :575-578: Linear below one pixel per sample */
    const int hold = trigger_mode == 4 || trigger_mode == 3;                           /* ZeroCrossing, EnvelopeHold */
    const double triggerSampleOffset = hold ? (v->window_size * 0.5 - (double)(int)(v->window_size * 0.5)) - 1.5
                                            : (trigger_mode == 1 ? sample_offset : 0.0);
    const double triggerCycleSamples = trigger_mode == 1 ? cycle_samples : 0.0;        /* calculateTriggeringOffset :241-248 */
    long bufferOffset;
    if (trigger_mode == 2) bufferOffset = (long)ceil(fmod((double)transport_position, v->window_size));   /* :588-592 */
    else if (hold) bufferOffset = (long)ceil(triggerSampleOffset);                     /* :593-597 */
    else {                                                                             /* :598-612 */
        const long cycleBuffers = interpolation == 3 ? 2 : 1;
        if (trigger_mode != 0) quantizedCycleSamples = (long)ceil(triggerCycleSamples);
        bufferOffset = roundedWindow + cycleBuffers * quantizedCycleSamples;
    }
    roundedWindow = roundedWindow > 2 ? roundedWindow : 2;                             /* :615 */
    size_t n = 0;
    if (interpolation == 2 || interpolation == 0) {                                    /* Linear, :707-741; None: the same list as points, :652-700 */
        long p = ((long)cursor - bufferOffset) % (long)size;                           /* eval.startFrom(-(bufferOffset + 0)) */
        if (p < 0) p += (long)size;
        const float endCondition = (float)(roundedWindow + quantizedCycleSamples);     /* :631 */
        for (float i = 0; i < endCondition; i += 1) {
            if (n < max_points) {
                xyz[n * 3] = i; xyz[n * 3 + 1] = wave_eval(memA, memB, eval_mode, p); xyz[n * 3 + 2] = 0;
                if (rgba) rgba[n] = colour_mem ? colour_mem[p] : key;
            }
            ++n;
            if (++p == (long)size) p = 0;
        }
        return n;
    }
    if (interpolation == 1) {                                                          /* Rectangular, :746-789 */
        long p = ((long)cursor - bufferOffset) % (long)size;
        if (p < 0) p += (long)size;
        const float endCondition = (float)(roundedWindow + quantizedCycleSamples);
        uint32_t oldColour = colour_mem ? colour_mem[p] : key;                         /* evaluator.evaluateColour() before the loop */
        for (float i = 0; i < endCondition; i += 1) {
            const float y = wave_eval(memA, memB, eval_mode, p);
            const uint32_t col = colour_mem ? colour_mem[p] : key;
            if (n + 1 < max_points) {
                xyz[n * 3] = i; xyz[n * 3 + 1] = y; xyz[n * 3 + 2] = 0;
                xyz[n * 3 + 3] = i + 1; xyz[n * 3 + 4] = y; xyz[n * 3 + 5] = 0;
                if (rgba) { rgba[n] = oldColour; rgba[n + 1] = col; }
            }
            oldColour = col;
            n += 2;
            if (++p == (long)size) p = 0;
        }
        return n;
    }
    /* Lanczos, :790-891 */
    double samplePos;
    if (trigger_mode == 2) samplePos = fmod((double)transport_position, v->window_size) - 1;   /* :798-801 */
    else if (hold) samplePos = triggerSampleOffset;                                     /* :802-805 */
    else samplePos = triggerCycleSamples * 2 + v->window_size - (trigger_mode == 1 ? triggerSampleOffset : 0.0);   /* :810 */
    if (trigger_mode == 0 || trigger_mode == 2) samplePos = ceil(samplePos);            /* :814-819 */
    const double inc = horizontalDelta / (v->rendering_scale * ((double)v->width - 1));
    double unitSpacePos = v->left;
    const double samplesPerPixel = 1.0 / pixelsPerSample;
    samplePos += -unitSpacePos / inc * samplesPerPixel;
    double currentSample = floor(samplePos);
    long p = ((long)cursor + (-(long)floor(samplePos) - KernelSize)) % (long)size;       /* eval.startFrom, :829 */
    if (p < 0) p += (long)size;
    float kernel[KernelBufferSize];
    uint32_t currentColour = 0, nextColour = 0;                                         /* ColourT{} */
    for (int i = 0; i < KernelBufferSize; ++i) {
        kernel[i] = wave_eval(memA, memB, eval_mode, p);
        currentColour = nextColour; if (colour_mem) nextColour = colour_mem[p];         /* get(), :836-843 */
        if (++p == (long)size) p = 0;
    }
    do {
        double delta = currentSample - samplePos;
        while (delta > 1) {
            samplePos += 1;
            delta -= 1;
            memmove(kernel, kernel + 1, sizeof(float) * (KernelBufferSize - 1));
            kernel[KernelBufferSize - 1] = wave_eval(memA, memB, eval_mode, p);
            currentColour = nextColour; if (colour_mem) nextColour = colour_mem[p];
            if (++p == (long)size) p = 0;
        }
        const double y = sgzo_lanczos_filter_f64(kernel, KernelBufferSize, (double)KernelSize + delta, KernelSize);
        if (n < max_points) {
            xyz[n * 3] = (float)unitSpacePos; xyz[n * 3 + 1] = (float)y; xyz[n * 3 + 2] = 0;
            if (colour_mem && rgba) {
                uint8_t a[4], b[4], o[4];
                memcpy(a, &currentColour, 4); memcpy(b, &nextColour, 4);
                sgzo_colour_lerp_f64(a, b, delta, o);
                memcpy(&rgba[n], o, 4);
            }
        }
        ++n;
        currentSample += samplesPerPixel;
        unitSpacePos += inc;
    } while (unitSpacePos < (v->right + inc));
    return n;
}

size_t sgzo_scope_wave_plot(const sgzo_scope_view *v, int trigger_mode, int interpolation, const float *memA, const float *memB,
                            int eval_mode, size_t size, size_t cursor, float *xyz, size_t max_points)
{
    return sgzo_scope_wave_plot_ex(v, trigger_mode, interpolation, memA, memB, eval_mode, size, cursor, 0.0, 0.0, NULL, xyz, NULL, max_points);
}

/* drawPolarPlot (VectorscopeRendering.cpp:500-746) over an AudioBufferView of the history ring: memory memL / memR of `size` samples
 * with write cursor `cursor` -- section 0 = [cursor, size), section 1 = [0, cursor) (getItIndex / getItRange; UNVERIFIED vs cpl).
 * The fade ramp is one running fp32 SIMD sum across both sections (:528-543, :592, :634).  fade_history selects the colour variant
 * (:637-746): rgb_out gets colour * vSampleFade (SIMD body) or colour * (fade + 1) (scalar tail); without it rgb_out = colour.
 * lanes = elements_of<V> (8 = AVX).  xyz: (X len, Y len, fade - 1). */
void sgzo_vector_polar_view(const float *memL, const float *memR, size_t size, size_t cursor, uint32_t lanes, int fade_history,
                            const float colour[3], float *xyz, float *rgb_out)
{
    const float cosineRotation = -0.70710678118654752440f, sineRotation = 0.70710678118654752440f;
    const long V = (long)lanes;
    const float fadePerSample = 1.0f / (float)size;
    const float incremental = fadePerSample * (float)V;
    float sampleFade[64], outFade[64];
    for (long l = 0; l < V; ++l) { outFade[l] = fadePerSample * (float)l; sampleFade[l] = outFade[l]; }
    size_t out = 0;
    for (int section = 0; section < 2; ++section) {
        const float *left = section == 0 ? memL + cursor : memL;
        const float *right = section == 0 ? memR + cursor : memR;
        const long sectionSamples = section == 0 ? (long)(size - cursor) : (long)cursor;
        long i = 0;
        for (; i < (sectionSamples - V); i += V) {
            for (long l = 0; l < V; ++l) {
                const float vl = left[i + l], vr = right[i + l];
                const float length = fmaxf(fabsf(vl), fabsf(vr));
                const float vY = vl * cosineRotation - vr * sineRotation;
                const float vX = vl * sineRotation + vr * cosineRotation;
                float angle = atanf(vX / vY);
                if (vl == 0.0f && vr == 0.0f) angle = 0.0f;
                outFade[l] = sampleFade[l] - 1.0f;
                xyz[out * 3 + 0] = sinf(angle) * length;
                xyz[out * 3 + 1] = cosf(angle) * length;
                xyz[out * 3 + 2] = outFade[l];
                if (rgb_out) for (int k = 0; k < 3; ++k) rgb_out[out * 3 + k] = fade_history ? colour[k] * sampleFade[l] : colour[k];
                ++out;
            }
            for (long l = 0; l < V; ++l) sampleFade[l] += incremental;
        }
        long remaining = 0;
        const float currentSampleFade = outFade[V - 1];
        for (; i < sectionSamples; ++i, ++remaining) {
            const float vl = left[i], vr = right[i];
            const float length = fmaxf(fabsf(vl), fabsf(vr));
            const float vY = vl * cosineRotation - vr * sineRotation;
            const float vX = vl * sineRotation + vr * cosineRotation;
            float angle = atanf(vX / vY);
            if (vl == 0.0f && vr == 0.0f) angle = 0.0f;
            const float currentFade = currentSampleFade - (float)remaining * fadePerSample;
            xyz[out * 3 + 0] = sinf(angle) * length;
            xyz[out * 3 + 1] = cosf(angle) * length;
            xyz[out * 3 + 2] = currentFade;
            if (rgb_out) for (int k = 0; k < 3; ++k) rgb_out[out * 3 + k] = fade_history ? colour[k] * (currentFade + 1) : colour[k];
            ++out;
        }
        for (long l = 0; l < V; ++l) sampleFade[l] += fadePerSample * (float)remaining;
    }
}
