/*
 * spectrum_stream.c -- the Spectrum view AS A STREAM: what the reference does with the blocks its host hands it, restated with the
 * block structure (and the quirks that come with it) intact.  TEST INFRASTRUCTURE ONLY (see sgz_oracle.h).
 *
 * spectrum.c frames a buffer ideally (frame f = samples [f hop, f hop + W)); this file follows the two threads of the reference:
 *
 *   audio thread    AudioDispatcher::dispatch (Source/Spectrum/SpectrumDSP.cpp:63-108) -> per pair
 *                   TransformPair::audioEntryPoint (Source/Spectrum/TransformDSP.inl:1165-1211) -> the BLOCK overload of
 *                   prepareTransform (:234-484) -> doTransform (:487-502) -> addAudioFrame (:1139-1148) ->
 *                   blendAndDispatchSpectrums (SpectrumDSP.cpp:111-206) for the frames the callback produced.
 *                   Quirk Q1: every frame of one callback is prepared from the UN-offset block pointer
 *                   (`buffer`, :1192, while `offset` advances at :1202) and from the history as it was before the callback.
 *                   Quirk Q2: with a history longer than the window, `sizeToStopAt = W - (stop + extraDiscarded)` (:245, :256-257)
 *                   subtracts the surplus a second time -- the frame is `extraDiscarded` samples short and zero-padded.
 *   render thread   Spectrum::vectorGLRendering, DisplayMode::LineGraph (Source/Spectrum/SpectrumRendering.cpp:617-635): once per
 *                   video frame and pair the WHOLE-RING overload of prepareTransform (:39-231) -> doTransform -> mapToLinearSpace
 *                   -> postProcessStdTransform (:1438), i.e. the peak-decay filters advance once per rendered frame; in this
 *                   mode the audio thread transforms nothing (:1167) and only keeps the resonators running (:1206-1209).
 *
 * The audio history is cpl::AudioStream's (absent submodule).  What Signalizer relies on, and what is restated: a per-channel
 * circular buffer of `audioHistorySize` samples (Spectrum.cpp:472-477 asks for exactly the window size) whose AudioBufferView
 * exposes two contiguous segments, oldest first (getItIndex / getItRange, AudioStream::bufferIndices == 2; TransformDSP.inl:65-68),
 * starting out as silence; a listener is called BEFORE the block enters the history ("the abstract timeline consists of the old data
 * in the audio stream, with the following audio presented in this function", :1187-1188).  UNVERIFIED vs cpl, like every other cpl
 * restatement in this oracle.
 */
#include "sgz_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* AudioBufferView over one channel of the history: memory of `size` samples, `cursor` = the next write position = the oldest sample */
typedef struct { const float *mem; size_t size, cursor; } view_t;
static size_t view_size(const view_t *v) { return v->size; }
static size_t view_it_range(const view_t *v, size_t indice) { return indice == 0 ? v->size - v->cursor : v->cursor; }
static const float *view_it_index(const view_t *v, size_t indice) { return indice == 0 ? v->mem + v->cursor : v->mem; }
#define BUFFER_INDICES 2                                     /* AudioStream::bufferIndices */

typedef struct {
    sgzo_cf *audio;          /* getAudioMemory<std::complex<T>>(transformSize + 1) */
    sgzo_cf *csp;            /* getWork: 2 P complex                                */
    sgzo_cf *states;         /* lineGraphs[k].states  [graph][P]                    */
    sgzo_cf *results;        /* lineGraphs[k].results [graph][P]                    */
    sgzo_cf *resonator;      /* cresonator state [2][V][P]                          */
    sgzo_cf *sfbuf;          /* frames of this callback, [count][P]                 */
    size_t sfCount, sfCap;
    size_t processedSamplesSinceLastFrame;
} pair_t;

struct sgzo_spectrum_stream {
    sgzo_spectrum_params p;
    uint32_t displayMode;    /* SGZO_DISPLAY_* */
    uint32_t N;
    size_t history;          /* audioHistorySize */
    float *ring;             /* [2C][history] */
    size_t cursor;
    float *window, *mapped, *slope, *gain, *work0, *work1;
    size_t workCap;
    float ratios[SGZO_NUM_SPEC_COLOURS + 1], weights[2 * SGZO_RES_MAX_TERMS];
    sgzo_cf *coeff;
    int V;
    double windowScale;
    pair_t *pairs;
    sgzo_cf *mappedOut;      /* optional tap: csp of every frame of the current call, [frame][pair][2P] */
    size_t mappedOutCap;     /* in frames */
};

sgzo_spectrum_stream *sgzo_stream_create(const sgzo_spectrum_params *p, uint32_t display_mode, size_t history)
{
    if (!p || p->axis_points < 2 || p->num_pairs == 0 || p->hop == 0) return NULL;
    sgzo_spectrum_stream *s = (sgzo_spectrum_stream *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->p = *p;
    s->displayMode = display_mode;
    const uint32_t W = p->window_size, P = p->axis_points, C = p->num_pairs;
    s->N = sgzo_transform_size(W);
    s->history = history ? history : W;
    s->ring = (float *)calloc((size_t)2 * C * s->history, sizeof(float));
    s->window = (float *)calloc(s->N, sizeof(float));
    s->mapped = (float *)malloc(sizeof(float) * P);
    s->slope = (float *)malloc(sizeof(float) * P);
    s->gain = (float *)malloc(sizeof(float) * P);
    s->coeff = (sgzo_cf *)malloc(sizeof(sgzo_cf) * P * (2 * SGZO_RES_MAX_TERMS - 1));
    s->windowScale = sgzo_window(p->window_type, p->window_symmetry, p->window_alpha, p->window_beta, W, s->window);
    sgzo_remap_frequencies(p, s->mapped);
    sgzo_slope_map(p, s->mapped, s->slope);
    sgzo_colour_ratios(p->ratios, s->ratios);
    s->V = 1;
    if (p->algorithm == SGZO_ALGO_RSNT) sgzo_resonator_map(p, s->mapped, s->coeff, s->gain, s->weights, &s->V);
    s->pairs = (pair_t *)calloc(C, sizeof(pair_t));
    for (uint32_t i = 0; i < C; ++i) {
        pair_t *pr = &s->pairs[i];
        pr->audio = (sgzo_cf *)calloc((size_t)s->N + 1, sizeof(sgzo_cf));
        pr->csp = (sgzo_cf *)calloc((size_t)P * 2, sizeof(sgzo_cf));
        pr->states = (sgzo_cf *)calloc((size_t)SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
        pr->results = (sgzo_cf *)calloc((size_t)SGZO_NUM_GRAPHS * P, sizeof(sgzo_cf));
        pr->resonator = (sgzo_cf *)calloc((size_t)2 * (2 * SGZO_RES_MAX_TERMS - 1) * P, sizeof(sgzo_cf));
    }
    return s;
}

void sgzo_stream_destroy(sgzo_spectrum_stream *s)
{
    if (!s) return;
    for (uint32_t i = 0; i < s->p.num_pairs; ++i) {
        pair_t *pr = &s->pairs[i];
        free(pr->audio); free(pr->csp); free(pr->states); free(pr->results); free(pr->resonator); free(pr->sfbuf);
    }
    free(s->pairs); free(s->ring); free(s->window); free(s->mapped); free(s->slope); free(s->gain); free(s->coeff);
    free(s->work0); free(s->work1); free(s->mappedOut);
    free(s);
}

/* one sample of the frame buffer: the expressions of prepareTransform's channel-mode switch, in the reference's evaluation order
 * (both overloads use the same ones: :59-216 and :263-470) */
static sgzo_cf prepared(uint32_t mode, float l, float r, float w)
{
    sgzo_cf z = {0, 0};
    switch (mode) {
    case SGZO_CH_LEFT: z.re = l * w; break;
    case SGZO_CH_RIGHT: z.re = r * w; break;
    case SGZO_CH_MERGE: z.re = (l + r) * w * 0.5f; break;
    case SGZO_CH_SIDE: z.re = (l - r) * w * 0.5f; break;
    case SGZO_CH_MIDSIDE: z.re = (l + r) * w * 0.5f; z.im = (l - r) * w * 0.5f; break;
    default: z.re = l * w; z.im = r * w; break;                                       /* Phase, Separate, Complex */
    }
    return z;
}

/* TransformPair<T>::prepareTransform(constant, views), the whole-ring overload, TransformDSP.inl:39-231 */
static int prepare_transform_ring(const sgzo_spectrum_stream *s, pair_t *pr, const view_t views[2])
{
    const size_t windowSize = s->p.window_size;
    if (view_size(&views[0]) != view_size(&views[1]) || view_size(&views[0]) < windowSize) return 0;     /* :45-46 */
    size_t offset = view_size(&views[0]) - windowSize;                                                   /* :49 */
    if (s->p.algorithm != SGZO_ALGO_FFT) return 1;                                                       /* switch (constant.algo): FFT only */
    sgzo_cf *buffer = pr->audio;
    size_t i = 0;
    for (size_t indice = 0; indice < BUFFER_INDICES; ++indice) {                                         /* e.g. :63-88 */
        size_t range = view_it_range(&views[0], indice);
        const float *left = view_it_index(&views[0], indice), *right = view_it_index(&views[1], indice);
        if (range > offset) {
            range -= offset; left += offset; right += offset;
            while (range--) {
                buffer[i] = prepared(s->p.channel_mode, *left++, *right++, s->window[i]);
                i++;
            }
            offset = 0;
        } else {
            offset -= range;
        }
    }
    for (size_t pad = i; pad < s->N; ++pad) { buffer[pad].re = 0; buffer[pad].im = 0; }                  /* :220-223 */
    return 1;
}

/* TransformPair<T>::prepareTransform(constant, views, preliminaryAudio, numSamples), the block overload, TransformDSP.inl:234-484 */
static int prepare_transform_block(const sgzo_spectrum_stream *s, pair_t *pr, const view_t views[2], const float *const preliminaryAudio[2],
                                   size_t numSamples)
{
    const size_t windowSize = s->p.window_size;
    if (view_size(&views[0]) != view_size(&views[1]) || view_size(&views[0]) < windowSize) return 0;     /* :241-242 */
    const size_t extraDiscardedSamples = view_size(&views[0]) - windowSize;                              /* :245 */
    if (s->p.algorithm != SGZO_ALGO_FFT) return 1;
    sgzo_cf *buffer = pr->audio;
    size_t i = 0;
    const size_t stop = numSamples < windowSize ? numSamples : windowSize;                               /* :253 */
    size_t offset = stop + extraDiscardedSamples;                                                        /* :255 */
    const size_t sizeToStopAt = windowSize - offset;                                                     /* :256 (Q2; size_t: wraps when offset > W) */
    for (size_t indice = 0; indice < BUFFER_INDICES; ++indice) {                                         /* e.g. :267-293 */
        size_t range = view_it_range(&views[0], indice);
        const float *left = view_it_index(&views[0], indice), *right = view_it_index(&views[1], indice);
        if (range > offset) {
            range -= offset; left += offset; right += offset;
            while (range-- && i < sizeToStopAt) {
                buffer[i] = prepared(s->p.channel_mode, *left++, *right++, s->window[i]);
                i++;
            }
            offset = 0;
        } else {
            offset -= range;
        }
    }
    for (size_t k = 0; k < stop; ++i, k++)                                                               /* "process preliminary", e.g. :297-300 */
        buffer[i] = prepared(s->p.channel_mode, preliminaryAudio[0][k], preliminaryAudio[1][k], s->window[i]);
    for (size_t pad = i; pad < s->N; ++pad) { buffer[pad].re = 0; buffer[pad].im = 0; }                  /* :473-476 */
    return 1;
}

static void tap_mapped(sgzo_spectrum_stream *s, size_t frame, uint32_t pairIndex, const sgzo_cf *csp)
{
    if (!s->mappedOut || frame >= s->mappedOutCap) return;
    const size_t P = s->p.axis_points;
    memcpy(s->mappedOut + (frame * s->p.num_pairs + pairIndex) * 2 * P, csp, sizeof(sgzo_cf) * 2 * P);
}

/* mapToLinearSpace (:506-1133) into pr->csp, whichever the algorithm */
static void map_to_linear_space(sgzo_spectrum_stream *s, pair_t *pr)
{
    const size_t P = s->p.axis_points;
    memset(pr->csp, 0, sizeof(sgzo_cf) * P * 2);
    if (s->p.algorithm == SGZO_ALGO_FFT) {
        sgzo_map_to_linear_space(&s->p, s->mapped, s->windowScale, pr->audio, s->N, pr->csp);
    } else {                                                                                             /* :1103-1133 */
        const int signals = s->p.channel_mode == SGZO_CH_LEFT || s->p.channel_mode == SGZO_CH_RIGHT || s->p.channel_mode == SGZO_CH_MERGE
                                    || s->p.channel_mode == SGZO_CH_SIDE ? 1 : 2;
        sgzo_resonator_windowed_state(&s->p, pr->resonator, s->gain, s->weights, s->V, signals, pr->csp);
    }
}

/* doTransform, :487-502 */
static void do_transform(sgzo_spectrum_stream *s, pair_t *pr)
{
    if (s->p.algorithm == SGZO_ALGO_FFT) {
        pr->audio[s->N].re = pr->audio[s->N].im = 0;              /* (mono modes never write csf[N]; defined as 0, as in spectrum.c) */
        sgzo_fft_forward(pr->audio, s->N);
    }
}

/* addAudioFrame, :1139-1148 */
static void add_audio_frame(sgzo_spectrum_stream *s, pair_t *pr, uint32_t pairIndex)
{
    const size_t P = s->p.axis_points;
    map_to_linear_space(s, pr);
    tap_mapped(s, pr->sfCount, pairIndex, pr->csp);
    sgzo_map_and_transform_filters(&s->p, s->slope, pr->csp, pr->states, pr->results);                   /* postProcessStdTransform, :1438 */
    if (pr->sfCount == pr->sfCap) {
        pr->sfCap = pr->sfCap ? pr->sfCap * 2 : 8;
        pr->sfbuf = (sgzo_cf *)realloc(pr->sfbuf, sizeof(sgzo_cf) * pr->sfCap * P);
    }
    memcpy(pr->sfbuf + pr->sfCount * P, pr->results, sizeof(sgzo_cf) * P);                               /* LineMain's results */
    pr->sfCount++;
}

/* resonatingDispatch, :1213-1295 */
static void resonating_dispatch(sgzo_spectrum_stream *s, pair_t *pr, const float *L, const float *R, size_t numSamples)
{
    if (numSamples < 1) return;                                                                          /* :1245-1246 */
    if (s->workCap < numSamples) {
        free(s->work0); free(s->work1);
        s->work0 = (float *)malloc(sizeof(float) * numSamples); s->work1 = (float *)malloc(sizeof(float) * numSamples);
        s->workCap = numSamples;
    }
    const float *work[2] = {s->work0, s->work1};
    const int signals = sgzo_resonator_dispatch(s->p.channel_mode, L, R, numSamples, s->work0, s->work1);
    sgzo_resonate_real(s->coeff, s->p.axis_points, s->V, pr->resonator, work, signals, numSamples);
}

/* TransformPair<T>::audioEntryPoint, :1165-1211 */
static void audio_entry_point(sgzo_spectrum_stream *s, pair_t *pr, uint32_t pairIndex, const view_t views[2], const float *const buffer[2],
                              size_t numSamples)
{
    const int64_t sampleBufferSize = (int64_t)s->p.hop;
    if (s->displayMode == SGZO_DISPLAY_COLOUR_SPECTRUM) {
        int64_t n = (int64_t)numSamples;
        size_t offset = 0;
        while (n > 0) {
            const int64_t psslf = (int64_t)pr->processedSamplesSinceLastFrame;
            const int64_t numRemainingSamples = psslf > sampleBufferSize ? 0 : sampleBufferSize - psslf;                 /* :1174 */
            const int64_t under = n - numRemainingSamples;
            const int64_t availableSamples = numRemainingSamples + (under < 0 ? under : 0);                               /* :1175 */
            if (s->p.algorithm == SGZO_ALGO_RSNT)
                resonating_dispatch(s, pr, buffer[0] + offset, buffer[1] + offset, (size_t)availableSamples);             /* :1178-1181 */
            pr->processedSamplesSinceLastFrame += (size_t)availableSamples;
            if ((int64_t)pr->processedSamplesSinceLastFrame >= sampleBufferSize) {
                int transformReady = 1;
                if (s->p.algorithm == SGZO_ALGO_FFT) {
                    /* Q1: `buffer`, not `buffer + offset` (:1192) */
                    if ((transformReady = prepare_transform_block(s, pr, views, buffer, (size_t)availableSamples)))
                        do_transform(s, pr);
                }
                if (transformReady) add_audio_frame(s, pr, pairIndex);
                pr->processedSamplesSinceLastFrame = 0;
            }
            offset += (size_t)availableSamples;
            n -= availableSamples;
        }
    } else if (s->p.algorithm == SGZO_ALGO_RSNT) {
        resonating_dispatch(s, pr, buffer[0], buffer[1], numSamples);                                                     /* :1206-1209 */
    }
}

static void views_of(const sgzo_spectrum_stream *s, uint32_t pairIndex, view_t views[2])
{
    for (int c = 0; c < 2; ++c) {
        views[c].mem = s->ring + ((size_t)2 * pairIndex + c) * s->history;
        views[c].size = s->history;
        views[c].cursor = s->cursor;
    }
}

/* onStreamAudio -> AudioDispatcher::dispatch (SpectrumDSP.cpp:63-108): every pair's audioEntryPoint on the same authority counter,
 * then blendAndDispatchSpectrums over the frames the callback produced; afterwards the block enters the history (cpl::AudioStream).
 * rgba_out [max_frames][P][4], line_out [max_frames][pair][graph][P], mapped_out [max_frames][pair][2P] (each optional).
 * Returns the number of frames the callback produced (those beyond max_frames are computed -- the states advance -- but not stored). */
long sgzo_stream_audio(sgzo_spectrum_stream *s, const float *const *planar, size_t n, uint8_t *rgba_out, sgzo_cf *line_out,
                       sgzo_cf *mapped_out, size_t max_frames)
{
    const uint32_t C = s->p.num_pairs;
    const size_t P = s->p.axis_points;
    if (mapped_out && s->mappedOutCap < max_frames) {
        free(s->mappedOut);
        s->mappedOut = (sgzo_cf *)malloc(sizeof(sgzo_cf) * max_frames * C * 2 * P);
        s->mappedOutCap = max_frames;
    }
    if (!mapped_out) { free(s->mappedOut); s->mappedOut = NULL; s->mappedOutCap = 0; }
    const size_t authorityCounter = s->pairs[0].processedSamplesSinceLastFrame;                          /* :81 */
    for (uint32_t i = 0; i < C; ++i) {
        pair_t *pr = &s->pairs[i];
        view_t views[2];
        views_of(s, i, views);
        const float *const buffer[2] = {planar[2 * i], planar[2 * i + 1]};
        pr->processedSamplesSinceLastFrame = authorityCounter;                                           /* :94 */
        pr->sfCount = 0;
        audio_entry_point(s, pr, i, views, buffer, n);
    }
    const size_t frames = s->pairs[0].sfCount;
    /* blendAndDispatchSpectrums (SpectrumDSP.cpp:111-206): frame s of every pair into one column */
    if (s->displayMode == SGZO_DISPLAY_COLOUR_SPECTRUM && frames > 0) {
        sgzo_cf *column = (sgzo_cf *)malloc(sizeof(sgzo_cf) * C * P);
        for (size_t f = 0; f < frames && f < max_frames; ++f) {
            for (uint32_t i = 0; i < C; ++i) memcpy(column + (size_t)i * P, s->pairs[i].sfbuf + f * P, sizeof(sgzo_cf) * P);
            if (rgba_out) sgzo_blend_column(&s->p, s->ratios, column, C, rgba_out + f * P * 4);
        }
        free(column);
    }
    if (mapped_out && s->mappedOut) memcpy(mapped_out, s->mappedOut, sizeof(sgzo_cf) * (frames < max_frames ? frames : max_frames) * C * 2 * P);
    if (line_out)                                                                                        /* the results after the callback's last frame */
        for (uint32_t i = 0; i < C; ++i)
            memcpy(line_out + (size_t)i * SGZO_NUM_GRAPHS * P, s->pairs[i].results, sizeof(sgzo_cf) * SGZO_NUM_GRAPHS * P);
    /* the block enters the history */
    for (size_t k = 0; k < n; ++k) {
        for (uint32_t c = 0; c < 2 * C; ++c) s->ring[(size_t)c * s->history + s->cursor] = planar[c][k];
        s->cursor = s->cursor + 1 == s->history ? 0 : s->cursor + 1;
    }
    return (long)frames;
}

/* Spectrum::vectorGLRendering, DisplayMode::LineGraph (SpectrumRendering.cpp:617-635): per pair prepareTransform(constant, views) ->
 * doTransform -> mapToLinearSpace -> postProcessStdTransform.  results [pair][graph][P] = lineGraphs[k].results, mapped_out (optional)
 * [pair][2P] = csp.  Returns 1, or 0 when prepareTransform refused (history shorter than the window). */
int sgzo_stream_render_lines(sgzo_spectrum_stream *s, sgzo_cf *results, sgzo_cf *mapped_out)
{
    const uint32_t C = s->p.num_pairs;
    const size_t P = s->p.axis_points;
    int all = 1;
    for (uint32_t i = 0; i < C; ++i) {
        pair_t *pr = &s->pairs[i];
        view_t views[2];
        views_of(s, i, views);
        if (prepare_transform_ring(s, pr, views)) {
            do_transform(s, pr);
            map_to_linear_space(s, pr);
            if (mapped_out) memcpy(mapped_out + (size_t)i * 2 * P, pr->csp, sizeof(sgzo_cf) * 2 * P);
            sgzo_map_and_transform_filters(&s->p, s->slope, pr->csp, pr->states, pr->results);
        } else all = 0;
        if (results) memcpy(results + (size_t)i * SGZO_NUM_GRAPHS * P, pr->results, sizeof(sgzo_cf) * SGZO_NUM_GRAPHS * P);
    }
    return all;
}

/* the GIVEN-values twin of the two entry points above for the parity chain: advance the filters of every pair by one frame of somebody
 * else's csp [pair][2P] (the device's own mapped pixels) and hand back results / the blended column */
void sgzo_stream_filters_given(sgzo_spectrum_stream *s, const sgzo_cf *csp_all, sgzo_cf *results, uint8_t *rgba_out)
{
    const uint32_t C = s->p.num_pairs;
    const size_t P = s->p.axis_points;
    sgzo_cf *column = rgba_out ? (sgzo_cf *)malloc(sizeof(sgzo_cf) * C * P) : NULL;
    for (uint32_t i = 0; i < C; ++i) {
        pair_t *pr = &s->pairs[i];
        sgzo_map_and_transform_filters(&s->p, s->slope, csp_all + (size_t)i * 2 * P, pr->states, pr->results);
        if (results) memcpy(results + (size_t)i * SGZO_NUM_GRAPHS * P, pr->results, sizeof(sgzo_cf) * SGZO_NUM_GRAPHS * P);
        if (column) memcpy(column + (size_t)i * P, pr->results, sizeof(sgzo_cf) * P);
    }
    if (column) { sgzo_blend_column(&s->p, s->ratios, column, C, rgba_out); free(column); }
}

/* the history as a frame firing now would see it: the `count` newest samples of channel `channel`, oldest first */
void sgzo_stream_history(const sgzo_spectrum_stream *s, uint32_t channel, size_t count, float *out)
{
    for (size_t k = 0; k < count; ++k)
        out[k] = s->ring[(size_t)channel * s->history + (s->cursor + s->history - count + k) % s->history];
}

size_t sgzo_stream_counter(const sgzo_spectrum_stream *s) { return s->pairs[0].processedSamplesSinceLastFrame; }
