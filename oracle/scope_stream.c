/*
 * scope_stream.c -- CPU restatement of the Oscilloscope's audio-thread state machine and of its render-thread peak filter.
 * TEST INFRASTRUCTURE (see sgz_oracle.h).  Follows:
 *   Source/Oscilloscope/OscilloscopeDSP.inl:401-424      StreamState::audioEntryPoint
 *   Source/Oscilloscope/StreamPreprocessing.h:46-76      TriggeringProcessor::setSettings / update
 *   Source/Oscilloscope/StreamPreprocessing.h:79-206     TriggeringProcessor::processMutating (trigger -> window -> front buffer)
 *   Source/Oscilloscope/StreamPreprocessing.h:315-349    ZeroCrossingProcessor::process (via executeSamplingWindows, OscilloscopeDSP.inl:311-385)
 *   Source/Oscilloscope/ChannelData.h:147-161            ChannelData::swapBuffers
 *   Source/Oscilloscope/OscilloscopeDSP.inl:427-710      StreamState::audioProcessing: RMS envelope (:520-585, :676-693), ring write (:696-697)
 *   Source/Oscilloscope/OscilloscopeDSP.inl:713-886      Oscilloscope::runPeakFilter, every OscChannels mode
 *   Source/Oscilloscope/OscilloscopeParameters.h:491-507 calculateTriggerIndices
 * The per-sample colours of audioProcessing (:445-517, :588-647: LR4 crossover -> band energies -> RGB) run through scope_spectral.c's
 * restatement of the cpl filters when sgzo_scope_stream_enable_colours was called; they live in rings parallel to the audio rings
 * (Channel::colourData / auxColourData, ChannelData.h:58-66) and are swapped with them (ChannelData.h:155-159).
 *
 * cpl::CLIFOStream<float> is restated as a ring of `size` elements with a write cursor (UNVERIFIED vs cpl, which is absent):
 *   createWriter().copyIntoHead(src, n)                      appends n samples at the cursor, wrapping at `size`;
 *   createWriter().copyIntoHead(view, historySize, offset)   appends historySize samples read from the other stream starting at
 *                                                            its cursor + offset (offset < 0: that many samples back in time);
 *   createProxyView(): begin() = the ring's memory, cursorPosition() = the write cursor = the oldest sample.
 */
#include "sgz_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { OSC_LEFT = 0, OSC_RIGHT = 1, OSC_MID = 2, OSC_SIDE = 3, OSC_SEPARATE = 4, OSC_MIDSIDE = 5 };
enum { TRIG_NONE = 0, TRIG_SPECTRAL = 1, TRIG_WINDOW = 2, TRIG_ENVELOPE_HOLD = 3, TRIG_ZERO_CROSSING = 4 };            /* OscilloscopeContent::TriggeringMode, OscilloscopeParameters.h:50-58 */
enum { ENV_NONE = 0, ENV_RMS = 1, ENV_PEAK_DECAY = 2 };    /* EnvelopeModes, CommonSignalizer.h */

typedef struct { float *buf; size_t size, cursor; } ring_t;

static void ring_write(ring_t *r, const float *src, size_t n)
{
    if (r->size == 0) return;
    for (size_t i = 0; i < n; ++i) {
        r->buf[r->cursor] = src[i];
        if (++r->cursor == r->size) r->cursor = 0;
    }
}

/* CLIFOStream<PixelType>: the same ring over RGBA8 pixels */
typedef struct { uint32_t *buf; size_t size, cursor; } cring_t;
static void cring_write(cring_t *r, const uint32_t *src, size_t n)
{
    if (r->size == 0) return;
    for (size_t i = 0; i < n; ++i) {
        r->buf[r->cursor] = src[i];
        if (++r->cursor == r->size) r->cursor = 0;
    }
}
static void cring_copy_head(cring_t *dst, const cring_t *src, size_t historySize, long offset)
{
    if (dst->size == 0 || src->size == 0) return;
    long rd = ((long)src->cursor + offset) % (long)src->size;
    if (rd < 0) rd += (long)src->size;
    for (size_t i = 0; i < historySize; ++i) {
        dst->buf[dst->cursor] = src->buf[rd];
        if (++dst->cursor == dst->size) dst->cursor = 0;
        if (++rd == (long)src->size) rd = 0;
    }
}

static void ring_copy_head(ring_t *dst, const ring_t *src, size_t historySize, long offset)
{
    if (dst->size == 0 || src->size == 0) return;
    long rd = ((long)src->cursor + offset) % (long)src->size;
    if (rd < 0) rd += (long)src->size;
    for (size_t i = 0; i < historySize; ++i) {
        dst->buf[dst->cursor] = src->buf[rd];
        if (++dst->cursor == dst->size) dst->cursor = 0;
        if (++rd == (long)src->size) rd = 0;
    }
}

/* std::queue<std::uint64_t> peaks */
typedef struct { uint64_t *v; size_t cap, head, count; } queue_t;
static void q_push(queue_t *q, uint64_t x)
{
    if (q->count == q->cap) {
        const size_t ncap = q->cap ? q->cap * 2 : 64;
        uint64_t *nv = (uint64_t *)malloc(sizeof(uint64_t) * ncap);
        for (size_t i = 0; i < q->count; ++i) nv[i] = q->v[(q->head + i) % q->cap];
        free(q->v); q->v = nv; q->cap = ncap; q->head = 0;
    }
    q->v[(q->head + q->count++) % q->cap] = x;
}
static uint64_t q_front(const queue_t *q) { return q->v[q->head]; }
static void q_pop(queue_t *q) { if (q->count) { q->head = (q->head + 1) % q->cap; --q->count; } }

struct sgzo_scope_stream {
    uint32_t channels, osc_mode, env_mode;
    int trigger_mode;
    size_t trigger_separate, trigger_pair;
    double sample_rate, envelope_window;
    ring_t *back, *front;                          /* ChannelData::back / front, audioData per channel */
    float *envelope;                               /* filterStates.channels[c].envelope */
    double envelope_gain;                          /* StreamState::envelopeGain */
    uint64_t playhead;                             /* ctx.getPlayhead().getSteadyClock(): samples delivered so far */
    /* TriggeringProcessor (value-initialised by make_unique<TriggeringProcessor>(): all zero) */
    double hysteresis, threshold, windowSize, state;
    int windowChanged, isPeakHold, isWorkingOnPeak;
    uint64_t crossOrigin, oldPeak, currentPeak, bufferedSamples, frontOrigin, steadyClock;
    queue_t peaks;
    uint64_t swaps;                                /* diagnostic: number of swapBuffers so far */
    /* frequency colouring (audioProcessing :445-517, :588-647) */
    int colours_on;
    sgzo_lr_coeffs network_coeffs;                 /* channelData.networkCoeffs */
    float smooth_pole;                             /* channelData.smoothFilterPole */
    float band_colours[3][3];                      /* content->lowColour / midColour / highColour as float r, g, b */
    float blend;                                   /* 1 - frequencyColouringBlend */
    sgzo_lr_state *network;                        /* filterStates.channels[c].network */
    float (*smooth)[3], (*aux_smooth)[3];          /* filterStates.channels[c].smoothFilters / auxSmoothFilter */
    uint8_t (*keys)[4];                            /* filterStates.channels[c].defaultKey as RGBA8 */
    cring_t *back_col, *back_aux, *front_col, *front_aux;
};

/* ChannelData::resizeAudioStorage (ChannelData.h:106-129).  Non-spectral modes: size = ceil(effectiveWindowSize + 1).  Spectral: the
 * reference resizes every frame to max((size_t)(0.5 + cycleSamples + ceil(window)), LookaheadSize); this stream keeps the largest
 * such ring (cycleSamples <= sampleRate / 5, the 5 Hz floor of calculateFundamentalPeriod) and sgzo_scope_stream_logical() cuts the
 * reference's ring of the moment out of it (the newest `size` samples, oldest first = a ring whose cursor is 0). */
static size_t storage_size(double window, int trigger_mode, double sample_rate)
{
    if (trigger_mode == TRIG_SPECTRAL) {
        const size_t n = (size_t)(0.5 + sample_rate / 5.0 + ceil(window));
        return n > 8192 ? n : 8192;
    }
    return (size_t)ceil(window + 1);
}

sgzo_scope_stream *sgzo_scope_stream_create(uint32_t channels, double sample_rate, double window_size, int trigger_mode,
                                            double threshold, uint32_t osc_mode, double trigger_channel_1based,
                                            uint32_t env_mode, double envelope_window_s)
{
    sgzo_scope_stream *s = (sgzo_scope_stream *)calloc(1, sizeof(*s));
    s->channels = channels; s->osc_mode = osc_mode; s->env_mode = env_mode; s->trigger_mode = trigger_mode;
    s->sample_rate = sample_rate; s->envelope_window = envelope_window_s;
    s->back = (ring_t *)calloc(channels, sizeof(ring_t));
    s->front = (ring_t *)calloc(channels, sizeof(ring_t));
    const size_t size = storage_size(window_size, trigger_mode, sample_rate);
    for (uint32_t c = 0; c < channels; ++c) {
        s->back[c].buf = (float *)calloc(size, sizeof(float)); s->back[c].size = size;
        s->front[c].buf = (float *)calloc(size, sizeof(float)); s->front[c].size = size;
    }
    s->envelope = (float *)calloc(channels, sizeof(float));
    /* calculateTriggerIndices, OscilloscopeParameters.h:491-507 (cpl::Math::round = nearest) */
    const size_t idx = (size_t)llround(trigger_channel_1based - 1);
    s->trigger_separate = idx < channels - 1 ? idx : channels - 1;
    s->trigger_pair = (idx < channels / 4 ? idx : channels / 4) * 2;
    /* TriggeringProcessor::setSettings, StreamPreprocessing.h:46-53 (hysteresis is unused by the zero-crossing detector) */
    s->windowChanged = ceil(window_size) != ceil(s->windowSize);
    s->windowSize = window_size;
    s->threshold = threshold;
    return s;
}

/* TriggeringProcessor::setSettings' hysteresis (StreamPreprocessing.h:46-53, Oscilloscope.cpp:310): only the envelope-hold detector reads it */
void sgzo_scope_stream_set_hysteresis(sgzo_scope_stream *s, double hysteresis) { s->hysteresis = hysteresis; }

void sgzo_scope_stream_destroy(sgzo_scope_stream *s)
{
    if (!s) return;
    for (uint32_t c = 0; c < s->channels; ++c) { free(s->back[c].buf); free(s->front[c].buf); }
    if (s->colours_on)
        for (uint32_t c = 0; c < s->channels; ++c) {
            free(s->back_col[c].buf); free(s->back_aux[c].buf); free(s->front_col[c].buf); free(s->front_aux[c].buf);
        }
    free(s->back_col); free(s->back_aux); free(s->front_col); free(s->front_aux);
    free(s->network); free(s->smooth); free(s->aux_smooth); free(s->keys);
    free(s->back); free(s->front); free(s->envelope); free(s->peaks.v); free(s);
}

/* switches the per-sample colours on (state.colourChannelsByFrequency only decides whether the renderer reads them; the reference
 * always computes them).  band_colours: low / mid / high as float r, g, b (ColourValue::getValue()); keys: RGBA8 per channel. */
void sgzo_scope_stream_enable_colours(sgzo_scope_stream *s, const float band_colours[3][3], double frequency_colouring_blend,
                                      double colour_smoothing_ms, const uint8_t (*keys)[4])
{
    const uint32_t C = s->channels;
    s->colours_on = 1;
    memcpy(s->band_colours, band_colours, sizeof(s->band_colours));
    s->blend = 1 - (float)frequency_colouring_blend;                                       /* :515 */
    s->smooth_pole = sgzo_colour_smooth_pole(colour_smoothing_ms, s->sample_rate);          /* tuneColourSmoothing, :439 */
    sgzo_lr_design(300, 3000, s->sample_rate, &s->network_coeffs);                          /* tuneCrossOver(300, 3000, sampleRate), :440 */
    s->network = (sgzo_lr_state *)calloc(C, sizeof(sgzo_lr_state));
    s->smooth = calloc(C, sizeof(float[3])); s->aux_smooth = calloc(C, sizeof(float[3]));
    s->keys = calloc(C, 4);
    memcpy(s->keys, keys, (size_t)C * 4);
    s->back_col = calloc(C, sizeof(cring_t)); s->back_aux = calloc(C, sizeof(cring_t));
    s->front_col = calloc(C, sizeof(cring_t)); s->front_aux = calloc(C, sizeof(cring_t));
    const size_t size = s->front[0].size;
    for (uint32_t c = 0; c < C; ++c) {
        cring_t *all[4] = {&s->back_col[c], &s->back_aux[c], &s->front_col[c], &s->front_aux[c]};
        for (int k = 0; k < 4; ++k) { all[k]->buf = (uint32_t *)calloc(size, sizeof(uint32_t)); all[k]->size = size; }
    }
}

static uint32_t pack_rgba(const uint8_t p[4]) { uint32_t v; memcpy(&v, p, 4); return v; }

/* the colour part of audioProcessing for numChannels >= 2 (:588-647): per channel pair, per sample */
static void colour_processing(sgzo_scope_stream *s, const float *const *buffer, size_t numSamples, int into_back)
{
    uint32_t *cl = (uint32_t *)malloc(sizeof(uint32_t) * numSamples * 4), *cr = cl + numSamples, *cm = cr + numSamples, *cs = cm + numSamples;
    for (uint32_t pair = 0; pair < s->channels; pair += 2) {
        sgzo_lr_state *netLeft = &s->network[pair], *netRight = &s->network[pair + 1];
        float *smLeft = s->smooth[pair], *smRight = s->smooth[pair + 1], *smMid = s->aux_smooth[pair], *smSide = s->aux_smooth[pair + 1];
        const uint8_t *leftColour = s->keys[pair], *rightColour = s->keys[pair + 1];
        for (size_t n = 0; n < numSamples; ++n) {
            float leftBands[3], rightBands[3], mid[3], side[3];
            uint8_t px[4];
            sgzo_lr_process(netLeft, &s->network_coeffs, buffer[pair][n], leftBands);
            sgzo_lr_process(netRight, &s->network_coeffs, buffer[pair + 1][n], rightBands);
            sgzo_colour_filter_states(leftBands, smLeft, s->smooth_pole);
            sgzo_colour_filter_states(rightBands, smRight, s->smooth_pole);
            for (int i = 0; i < 3; ++i) { mid[i] = leftBands[i] + rightBands[i]; side[i] = leftBands[i] - rightBands[i]; }
            sgzo_colour_filter_states(mid, smMid, s->smooth_pole);
            sgzo_colour_filter_states(side, smSide, s->smooth_pole);
            sgzo_colour_accumulate(smLeft, s->band_colours, leftColour, s->blend, px); cl[n] = pack_rgba(px);
            sgzo_colour_accumulate(smRight, s->band_colours, rightColour, s->blend, px); cr[n] = pack_rgba(px);
            sgzo_colour_accumulate(smMid, s->band_colours, leftColour, s->blend, px); cm[n] = pack_rgba(px);
            sgzo_colour_accumulate(smSide, s->band_colours, rightColour, s->blend, px); cs[n] = pack_rgba(px);
        }
        cring_t *col = into_back ? s->back_col : s->front_col, *aux = into_back ? s->back_aux : s->front_aux;
        cring_write(&col[pair], cl, numSamples); cring_write(&col[pair + 1], cr, numSamples);      /* cwLeft / cwRight */
        cring_write(&aux[pair], cm, numSamples); cring_write(&aux[pair + 1], cs, numSamples);      /* cwMid / cwSide */
    }
    free(cl);
}

/* the part of StreamState::audioProcessing that is on this row: RMS envelope (:520-585, :676-693) and the ring write (:696-697) */
static void audio_processing(sgzo_scope_stream *s, const float *const *buffer, size_t numSamples, ring_t *target)
{
    const uint32_t numChannels = s->channels;
    if (target[0].size < 1) return;                                                    /* :443 */
    const float envelopeCoeff = (float)exp(-1.0 / (s->envelope_window * s->sample_rate)); /* :448 */
    float filterEnv[64];
    for (uint32_t i = 0; i < numChannels; ++i) filterEnv[i] = s->envelope[i];          /* :452-453 */
    uint32_t mode = s->osc_mode;
    if (numChannels >= 2 && s->env_mode != ENV_NONE) {
        size_t offset = 0;
        switch (mode) {
        case OSC_RIGHT: offset = 1; /* fall through */
        case OSC_LEFT:
            for (size_t n = 0; n < numSamples; ++n) {
                const float sample = buffer[offset][n] * buffer[offset][n];
                filterEnv[0] = sample + envelopeCoeff * (filterEnv[0] - sample);
            }
            for (uint32_t c = 1; c < numChannels; ++c) filterEnv[c] = filterEnv[0];
            break;
        case OSC_MID:
            for (size_t n = 0; n < numSamples; ++n) {
                const float mid = 0.5f * (buffer[0][n] + buffer[1][n]);
                const float sample = mid * mid;
                filterEnv[0] = sample + envelopeCoeff * (filterEnv[0] - sample);
            }
            for (uint32_t c = 1; c < numChannels; ++c) filterEnv[c] = filterEnv[0];
            break;
        case OSC_SIDE:
            for (size_t n = 0; n < numSamples; ++n) {
                const float side = 0.5f * (buffer[0][n] - buffer[1][n]);
                const float sample = side * side;
                filterEnv[0] = sample + envelopeCoeff * (filterEnv[0] - sample);
            }
            for (uint32_t c = 1; c < numChannels; ++c) filterEnv[c] = filterEnv[0];
            break;
        case OSC_SEPARATE:
            for (size_t n = 0; n < numSamples; ++n)
                for (uint32_t c = 0; c < numChannels; ++c) {
                    const float sample = buffer[c][n] * buffer[c][n];
                    filterEnv[c] = sample + envelopeCoeff * (filterEnv[c] - sample);
                }
            break;
        case OSC_MIDSIDE:
            for (size_t n = 0; n < numSamples; ++n) {
                const float left = buffer[0][n], right = buffer[1][n];
                const float mid = 0.5f * ((left + right) * (left + right)), side = 0.5f * ((left - right) * (left - right));
                filterEnv[0] = mid + envelopeCoeff * (filterEnv[0] - mid);
                filterEnv[1] = side + envelopeCoeff * (filterEnv[1] - side);
            }
            for (uint32_t c = 2; c < numChannels; ++c) filterEnv[c] = filterEnv[1];
            break;
        default: break;
        }
    }
    if (s->env_mode == ENV_RMS) {                                                      /* :676-693 */
        float start = sqrtf(filterEnv[0]);
        for (uint32_t c = 0; c < numChannels; ++c) start = fmaxf(start, sqrtf(filterEnv[c]));
        s->envelope_gain = 1.0 / start;
        s->envelope[0] = filterEnv[0];                                                 /* only Left and Right are stored back */
        s->envelope[1] = filterEnv[1];
    }
    if (s->colours_on) colour_processing(s, buffer, numSamples, target == s->back);
    for (uint32_t c = 0; c < numChannels; ++c) ring_write(&target[c], buffer[c], numSamples);   /* :696-697 */
}

/* executeSamplingWindows<ZeroCrossingProcessor | PeakHoldProcessor>, OscilloscopeDSP.inl:311-385 + StreamPreprocessing.h:270-349 */
static void pre_analyse(sgzo_scope_stream *s, const float *const *buffer, size_t numSamples)
{
    if (s->channels < 2) return;
    uint32_t localMode = s->osc_mode;
    size_t triggerPair = s->trigger_pair;
    if (localMode == OSC_MIDSIDE) { localMode = OSC_MID; triggerPair = s->trigger_separate & ~(size_t)1; }   /* :340-352 */
    const float *a, *b;
    switch (localMode) {
    case OSC_RIGHT: a = b = buffer[triggerPair + 1]; break;
    case OSC_LEFT: a = b = buffer[triggerPair]; break;
    case OSC_SEPARATE: a = b = buffer[s->trigger_separate]; break;
    default: a = buffer[triggerPair]; b = buffer[triggerPair + 1]; break;
    }
    size_t count = 0;
    for (size_t n = 0; n < numSamples; ++n) {
        double sample;
        if (localMode == OSC_MID) sample = (double)(0.5f * (a[n] + b[n]));
        else if (localMode == OSC_SIDE) sample = (double)(0.5f * (a[n] - b[n]));
        else sample = (double)a[n];
        if (s->trigger_mode == TRIG_ENVELOPE_HOLD) {                                    /* PeakHoldProcessor::process, :282-310 */
            sample *= sample;
            const double delta = sample - s->state;
            if (delta < 0) {
                s->state *= 0.9999;
                s->state = fmax(s->threshold * s->threshold, s->state);
                if (s->isPeakHold) {
                    /* minus one since this is the first sample that doesn't qualify as a (rising) peak */
                    q_push(&s->peaks, s->playhead + count - 1);
                    s->isPeakHold = 0;
                }
            } else {
                if (delta > s->hysteresis * s->state) s->isPeakHold = 1;
                s->state = sample;
            }
            count++;
            continue;
        }
        if (sample > 0 && s->state < 0) { s->isPeakHold = 1; s->crossOrigin = s->playhead + count; }
        if (s->isPeakHold && sample > s->threshold) { s->isPeakHold = 0; q_push(&s->peaks, s->crossOrigin); }
        s->state = sample;
        count++;
    }
}

static uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }

/* TriggeringProcessor::processMutating, StreamPreprocessing.h:79-206.  Integer / index work: every conversion between
 * std::uint64_t and double follows the reference's expression types. */
static void process_mutating(sgzo_scope_stream *s, const float **localPointers, size_t numSamples)
{
    if (s->frontOrigin + s->bufferedSamples < s->steadyClock) {                        /* :81-85 */
        s->frontOrigin = s->steadyClock;
        s->bufferedSamples = 0;
    }
    const double ceilingSize = ceil(s->windowSize);
    const double halfSize = ceilingSize / 2;
#define PROCESS_INTO_BACK(samples_)                                                                  \
    do {                                                                                             \
        const uint64_t smp = (uint64_t)(samples_);                                                   \
        audio_processing(s, localPointers, (size_t)smp, s->back);                                    \
        numSamples -= (size_t)smp;                                                                   \
        for (uint32_t c = 0; c < s->channels; ++c) localPointers[c] += smp;                          \
        const uint64_t oldSamples = s->bufferedSamples;                                              \
        s->steadyClock += smp;                                                                       \
        s->bufferedSamples += smp;                                                                   \
        s->bufferedSamples = min_u64(s->bufferedSamples, (uint64_t)(ceilingSize + 1));               \
        s->frontOrigin += (oldSamples + smp) - s->bufferedSamples;                                   \
    } while (0)

    if (ceilingSize == 0 && s->peaks.count) s->peaks.count = 0;                        /* :107-110 */
    while (numSamples != 0) {
        if (!s->peaks.count) {
            PROCESS_INTO_BACK(numSamples);
            break;
        } else if (!s->isWorkingOnPeak) {
            s->isWorkingOnPeak = 1;
            const uint64_t nextPeak = q_front(&s->peaks);
            if (nextPeak >= s->steadyClock) {
                const uint64_t deltaToPeak = nextPeak - s->steadyClock;
                const uint64_t numSamplesToProcess = min_u64((uint64_t)numSamples, (uint64_t)((double)deltaToPeak + halfSize));
                PROCESS_INTO_BACK(numSamplesToProcess);
                s->currentPeak = nextPeak;
            } else {
                s->currentPeak = nextPeak;
            }
        }
        uint64_t windowEnd;
        int isPeakOutsideOfWindow = 0, readyForBufferSwap = 0;
        if ((double)(s->currentPeak - s->oldPeak) < halfSize) {                        /* unsigned difference, :147 */
            windowEnd = (uint64_t)((double)s->oldPeak + halfSize);
        } else {
            isPeakOutsideOfWindow = 1;
            windowEnd = s->frontOrigin + s->bufferedSamples;
        }
        const uint64_t peakWindowEnd = (uint64_t)((double)((uint64_t)(isPeakOutsideOfWindow ? 1 : 0) + s->currentPeak) + halfSize);
        uint64_t numSamplesToProcess = 0;
        const uint64_t missingBufferSamples = peakWindowEnd - min_u64(peakWindowEnd, windowEnd);
        const uint64_t neededPreSamples =
            min_u64((uint64_t)halfSize, (uint64_t)(fmax((double)(s->currentPeak - s->oldPeak), halfSize) - halfSize));
        if (isPeakOutsideOfWindow) {
            numSamplesToProcess = min_u64((uint64_t)numSamples, missingBufferSamples);
            if (numSamplesToProcess > 0) PROCESS_INTO_BACK(numSamplesToProcess);
            readyForBufferSwap = missingBufferSamples == numSamplesToProcess;
        } else {
            if (s->bufferedSamples >= missingBufferSamples) {
                readyForBufferSwap = 1;
            } else {
                const uint64_t numRemaining = missingBufferSamples - s->bufferedSamples;
                numSamplesToProcess = min_u64((uint64_t)numSamples, numRemaining);
                if (numSamplesToProcess > 0) PROCESS_INTO_BACK(numSamplesToProcess);
                readyForBufferSwap = numRemaining == numSamplesToProcess;
            }
        }
        if (readyForBufferSwap) {
            const double amount = (isPeakOutsideOfWindow ? halfSize : (double)missingBufferSamples) + (double)neededPreSamples;
            const size_t cappedSize = (size_t)min_u64(s->bufferedSamples, (uint64_t)ceil(amount + 1));
            /* ChannelData::swapBuffers(cappedSize, -bufferedSamples), ChannelData.h:147-161 */
            for (uint32_t c = 0; c < s->channels; ++c) {
                ring_copy_head(&s->front[c], &s->back[c], cappedSize, -(long)s->bufferedSamples);
                if (s->colours_on) {
                    cring_copy_head(&s->front_col[c], &s->back_col[c], cappedSize, -(long)s->bufferedSamples);
                    cring_copy_head(&s->front_aux[c], &s->back_aux[c], cappedSize, -(long)s->bufferedSamples);
                }
            }
            s->bufferedSamples -= min_u64(s->bufferedSamples, (uint64_t)cappedSize);
            s->frontOrigin += cappedSize;
            s->oldPeak = s->currentPeak;
            s->isWorkingOnPeak = 0;
            q_pop(&s->peaks);
            s->swaps++;
        }
    }
#undef PROCESS_INTO_BACK
}

/* StreamState::audioEntryPoint, OscilloscopeDSP.inl:401-424.  planar: numChannels pointers to n samples. */
void sgzo_scope_stream_audio(sgzo_scope_stream *s, const float *const *planar, size_t n)
{
    if (n == 0 || s->channels == 0) return;
    const float *local[64];
    for (uint32_t c = 0; c < s->channels; ++c) local[c] = planar[c];
    /* TriggeringProcessor::update, StreamPreprocessing.h:55-76 */
    s->steadyClock = s->playhead;
    if (s->windowChanged) {
        s->windowChanged = 0;
        if (s->isWorkingOnPeak) q_pop(&s->peaks);
        while (s->peaks.count && q_front(&s->peaks) < s->playhead) q_pop(&s->peaks);
        s->bufferedSamples = s->currentPeak = s->oldPeak = 0;
        s->frontOrigin = s->playhead;
        s->isWorkingOnPeak = 0;
    }
    const int hold = s->trigger_mode == TRIG_ZERO_CROSSING || s->trigger_mode == TRIG_ENVELOPE_HOLD;
    if (hold) pre_analyse(s, local, n);                                                 /* preAnalyseAudio, :387-399 */
    if (!hold) audio_processing(s, local, n, s->front);                                 /* :415-418 */
    else process_mutating(s, local, n);                                                 /* :420-422 */
    s->playhead += n;
}

/* front buffer of channel c: raw ring memory (begin()) into out[size]; returns the write cursor (cursorPosition()) */
size_t sgzo_scope_stream_front(const sgzo_scope_stream *s, uint32_t c, float *out)
{
    memcpy(out, s->front[c].buf, sizeof(float) * s->front[c].size);
    return s->front[c].cursor;
}
/* front colour ring of channel c (aux = 0: colourData, 1: auxColourData): raw ring memory as RGBA8 words */
size_t sgzo_scope_stream_front_colours(const sgzo_scope_stream *s, uint32_t c, int aux, uint32_t *out)
{
    const cring_t *r = aux ? &s->front_aux[c] : &s->front_col[c];
    memcpy(out, r->buf, sizeof(uint32_t) * r->size);
    return r->cursor;
}
size_t sgzo_scope_stream_size(const sgzo_scope_stream *s) { return s->front[0].size; }
double sgzo_scope_stream_envelope_gain(const sgzo_scope_stream *s) { return s->envelope_gain; }
void sgzo_scope_stream_envelopes(const sgzo_scope_stream *s, float *out) { memcpy(out, s->envelope, sizeof(float) * s->channels); }
/* diagnostics: {frontOrigin, bufferedSamples, oldPeak, currentPeak, steadyClock, peaks.size(), isWorkingOnPeak, swaps} */
void sgzo_scope_stream_state(const sgzo_scope_stream *s, uint64_t out[8])
{
    out[0] = s->frontOrigin; out[1] = s->bufferedSamples; out[2] = s->oldPeak; out[3] = s->currentPeak;
    out[4] = s->steadyClock; out[5] = s->peaks.count; out[6] = (uint64_t)s->isWorkingOnPeak; out[7] = s->swaps;
}

/* Oscilloscope::runPeakFilter, OscilloscopeDSP.inl:713-886, on the stream's front buffers (raw ring memory from begin(), the last
 * n mod lanes memory slots dropped: SURVEY Q8).  coeff = pow(exp(-lanes / (envelopeWindow * sampleRate)), numSamples * dt) (:745-747)
 * is the caller's.  Returns state.autoGain = 1 / max_c sqrt(envelope_c). */
double sgzo_scope_stream_peak_filter(sgzo_scope_stream *s, uint32_t lanes, double coeff)
{
    const uint32_t numChannels = s->channels;
    const uint32_t channelMode = numChannels == 1 ? OSC_LEFT : s->osc_mode;
    const size_t numSamples = s->front[0].size;
    const size_t stop = numSamples - (numSamples & (size_t)(lanes - 1));
    float lMax = 0.0f, rMax = 0.0f;                        /* max over the SIMD lanes of vLMax / vRMax */
    if (channelMode <= OSC_SIDE) {                          /* OscChannels::OffsetForMono */
        const float *leftBuffer = s->front[0].buf;
        const float *rightBuffer = numChannels > 1 ? s->front[1].buf : leftBuffer;
        for (size_t i = 0; i < stop; ++i) {
            float v;
            switch (channelMode) {
            case OSC_LEFT: v = leftBuffer[i]; break;
            case OSC_RIGHT: v = rightBuffer[i]; break;
            case OSC_MID: v = (leftBuffer[i] + rightBuffer[i]) * 0.5f; break;
            default: v = (leftBuffer[i] - rightBuffer[i]) * 0.5f; break;
            }
            lMax = fmaxf(fabsf(v), lMax);
        }
        rMax = lMax;
        const double highestLeft = lMax, highestRight = rMax;
        s->envelope[0] = (float)fmax((double)s->envelope[0] * coeff, highestLeft * highestLeft);
        if (numChannels > 1) s->envelope[1] = (float)fmax((double)s->envelope[1] * coeff, highestRight * highestRight);
        for (uint32_t c = 2; c < numChannels; ++c) s->envelope[c] = s->envelope[1];
    } else if (channelMode == OSC_SEPARATE) {
        for (uint32_t c = 0; c < numChannels; ++c) {
            const float *buffer = s->front[c].buf;
            for (size_t i = 0; i < stop; ++i) lMax = fmaxf(fabsf(buffer[i]), lMax);    /* vLMax is NOT reset between channels (:827-838) */
            const float highestValue = lMax;
            /* std::max<float>(envelope * coeff, highestValue * highestValue): the double product is converted to float first */
            s->envelope[c] = fmaxf((float)((double)s->envelope[c] * coeff), highestValue * highestValue);
        }
    } else {                                                /* MidSide, :845-876 */
        const float *leftBuffer = s->front[0].buf, *rightBuffer = s->front[1].buf;
        for (size_t i = 0; i < stop; ++i) {
            const float a = leftBuffer[i] + rightBuffer[i], b = leftBuffer[i] - rightBuffer[i];
            lMax = fmaxf(fabsf(a * 0.5f), lMax);
            rMax = fmaxf(fabsf(b * 0.5f), rMax);
        }
        const double highestLeft = lMax, highestRight = rMax;
        s->envelope[0] = (float)fmax((double)s->envelope[0] * coeff, highestLeft * highestLeft);
        s->envelope[1] = (float)fmax((double)s->envelope[1] * coeff, highestRight * highestRight);
        for (uint32_t c = 2; c < numChannels; ++c) s->envelope[c] = s->envelope[1];
    }
    float start = sqrtf(s->envelope[0]);
    for (uint32_t c = 0; c < numChannels; ++c) start = fmaxf(start, sqrtf(s->envelope[c]));
    return 1.0 / start;
}
