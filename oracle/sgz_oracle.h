/*
 * sgz_oracle.h -- CPU restatement ("oracle") of Signalizer's visualiser DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The product (signalizer_amd/, libsgz.so) never
 * links, imports or falls back to it.
 *
 * PARITY UNPINNED.  The reference (jthorborg/signalizer v0.4.3) ships no tests, golden vectors
 * or fixtures, and every arithmetic primitive on the path lives in the un-vendored submodule
 * `cpl` (https://bitbucket.org/Mayae/cpl, pinned commit unrecoverable: /root/reference/External/cpl
 * is empty, .gitmodules:1-3).  The reference cannot be compiled here.  This oracle therefore
 * restates (a) the Signalizer-owned code in /root/reference/Source line by line (file:line cited
 * at each function) and (b) the cpl primitives from their published definitions, each such
 * primitive tagged "UNVERIFIED vs cpl".  It is pinned only against independent mathematics
 * (numpy fp64 FFT, scipy windows, closed-form known answers; see tests/test_oracle_*.py).
 *
 * Build: strict IEEE fp32/fp64, no contraction (-ffp-contract=off), see oracle/Makefile.
 */
#ifndef SGZ_ORACLE_H
#define SGZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } sgzo_cf;
typedef struct { double re, im; } sgzo_cd;

/* Source/Common/CommonSignalizer.h:495-539 (SpectrumChannels) */
enum {
    SGZO_CH_LEFT = 0, SGZO_CH_RIGHT = 1, SGZO_CH_MERGE = 2, SGZO_CH_SIDE = 3,
    SGZO_CH_PHASE = 4, SGZO_CH_SEPARATE = 5, SGZO_CH_MIDSIDE = 6, SGZO_CH_COMPLEX = 7
};
/* SpectrumContent::BinInterpolation (Source/Spectrum/SpectrumParameters.h) */
enum { SGZO_INTERP_NONE = 0, SGZO_INTERP_LINEAR = 1, SGZO_INTERP_LANCZOS = 2 };
/* SpectrumContent::ViewScaling */
enum { SGZO_VIEW_LINEAR = 0, SGZO_VIEW_LOG = 1 };
/* window shapes (cpl::dsp::WindowTypes is in the missing submodule; this is the build's own list) */
enum {
    SGZO_WIN_RECT = 0, SGZO_WIN_HANN = 1, SGZO_WIN_HAMMING = 2, SGZO_WIN_FLATTOP = 3,
    SGZO_WIN_BLACKMAN = 4, SGZO_WIN_EXACT_BLACKMAN = 5, SGZO_WIN_NUTTALL = 6,
    SGZO_WIN_BLACKMAN_NUTTALL = 7, SGZO_WIN_BLACKMAN_HARRIS = 8, SGZO_WIN_TRIANGULAR = 9,
    SGZO_WIN_WELCH = 10, SGZO_WIN_GAUSSIAN = 11, SGZO_WIN_KAISER = 12, SGZO_WIN_END = 13
};
enum { SGZO_WIN_SYMMETRIC = 0, SGZO_WIN_PERIODIC = 1 };

#define SGZO_NUM_SPEC_COLOURS 5
#define SGZO_NUM_GRAPHS 2

/* POD mirror of Signalizer::TransformConstant<float> (Source/Spectrum/TransformConstant.h:48-241)
 * plus the parameters Spectrum::handleFlagUpdates derives it from (Source/Spectrum/Spectrum.cpp:351-616). */
typedef struct sgzo_spectrum_params {
    float    sample_rate;        /* TransformConstant::sampleRate (T = float)                 */
    uint32_t window_size;        /* W, TransformConstant::windowSize                          */
    uint32_t hop;                /* sampleBufferSize (SpectrumDSP.cpp:51-54)                  */
    uint32_t axis_points;        /* P                                                         */
    uint32_t channel_mode;       /* SGZO_CH_*                                                 */
    uint32_t bin_interp;         /* SGZO_INTERP_*                                             */
    uint32_t view_scaling;       /* SGZO_VIEW_*                                               */
    uint32_t window_type;        /* SGZO_WIN_*                                                */
    uint32_t window_symmetry;    /* SGZO_WIN_SYMMETRIC / PERIODIC                             */
    uint32_t num_pairs;          /* stereo pairs blended into one column                      */
    double   window_alpha, window_beta;
    double   view_left, view_right;   /* state.viewRect                                       */
    double   min_log_freq;            /* state.minLogFreq (10 Hz, Spectrum.cpp:83)            */
    double   low_db, high_db, clip_db;/* TransformConstant::lowDBs/highDBs/clipDB             */
    double   slope_a, slope_b;        /* PowerSlopeValue::PowerFunction                       */
    float    pole[SGZO_NUM_GRAPHS];   /* TransformConstant::filter[k].pole                    */
    uint8_t  colours[SGZO_NUM_SPEC_COLOURS + 1][3]; /* [0]=background, 1..5 gradient, RGB8    */
    uint8_t  _pad[2];
    double   ratios[SGZO_NUM_SPEC_COLOURS];         /* content->specRatios normalised values  */
    uint32_t algorithm;          /* SGZO_ALGO_*: SpectrumContent::TransformAlgorithm (SpectrumParameters.h:66-69) */
    uint32_t free_q;             /* content->freeQ (Spectrum.cpp:593): RSNT bandwidths not bounded by the window  */
} sgzo_spectrum_params;
enum { SGZO_ALGO_FFT = 0, SGZO_ALGO_RSNT = 1 };
#define SGZO_RES_MAX_TERMS 5     /* cosine-sum terms of the widest window (flat top): 2 * 5 - 1 = 9 vectors */

/* ---------------- primitives (cpl restatements, UNVERIFIED vs cpl) ---------------- */
double   sgzo_window(uint32_t type, uint32_t symmetry, double alpha, double beta,
                     uint32_t W, float *out);            /* returns windowKernelScale */
uint32_t sgzo_transform_size(uint32_t W);               /* TransformConstant.h:84 */
void     sgzo_fft_forward(sgzo_cf *buf, uint32_t N);    /* fp32, unnormalised, natural order */
void     sgzo_fft_forward_f64(sgzo_cd *buf, uint32_t N);
void     sgzo_separate_transforms_ipl(sgzo_cf *csf, uint32_t N);
sgzo_cf  sgzo_lanczos_filter_wrap(const sgzo_cf *v, size_t size, float x, int a);
sgzo_cf  sgzo_linear_filter(const sgzo_cf *v, size_t size, float x);
double   sgzo_lanczos_filter_f64(const float *v, size_t size, double x, int a);
double   sgzo_lanczos_kernel(double d, int a);

/* ---------------- TransformConstant (a8) ---------------- */
void sgzo_remap_frequencies(const sgzo_spectrum_params *p, float *mapped /*P*/);
void sgzo_slope_map(const sgzo_spectrum_params *p, const float *mapped, float *slope /*P*/);
void sgzo_colour_ratios(const double ratios[SGZO_NUM_SPEC_COLOURS], float out[SGZO_NUM_SPEC_COLOURS + 1]);
void sgzo_rotate_hue_rgb8(const uint8_t rgb[3], float amount, uint8_t out[3]);
void sgzo_colour_table(const sgzo_spectrum_params *p, uint32_t pair, float sca[SGZO_NUM_SPEC_COLOURS + 1][3]);

/* ---------------- per-frame chain (a2..a7) ---------------- */
/* a2: Source/Spectrum/TransformDSP.inl:39-231; L,R: the W newest samples in time order */
void sgzo_prepare_transform(uint32_t mode, const float *L, const float *R, const float *window,
                            uint32_t W, uint32_t N, sgzo_cf *buf /*N+1*/);
/* a4: TransformDSP.inl:506-1102; csf is N+1 long and is modified in place; csp is 2*P complex */
int  sgzo_map_to_linear_space(const sgzo_spectrum_params *p, const float *mapped, double window_scale,
                              sgzo_cf *csf, uint32_t N, sgzo_cf *csp);
/* a5: TransformDSP.inl:1299-1435; states/results: [graph][P] of (float,float) */
void sgzo_map_and_transform_filters(const sgzo_spectrum_params *p, const float *slope,
                                    const sgzo_cf *csp, sgzo_cf *states, sgzo_cf *results);
/* a7: SpectrumDSP.cpp:111-206; frames: [pair][P] (magnitude,phase); out RGBA8 [P][4] */
void sgzo_blend_column(const sgzo_spectrum_params *p, const float *norm_ratios,
                       const sgzo_cf *frames, uint32_t num_pairs, uint8_t *rgba);

/* ---------------- whole offline job (ideal STFT framing) ---------------- */
/* planar: [2*num_pairs][nsamples]; rgba_out: [F][P][4]; line_out (optional): [F][pair][graph][P] complex;
 * mapped_out (optional): [F][pair][2*P] complex (csp) ; returns number of frames F (or <0 on error). */
long sgzo_spectrogram(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                      uint8_t *rgba_out, sgzo_cf *line_out, sgzo_cf *mapped_out);
/* same but only the FFT-heavy part of frames [f0,f1) - used by bench.py's bounded cpu_baseline sample */
long sgzo_spectrogram_range(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                            long f0, long f1, uint8_t *rgba_out);
/* timing only (bench.py cpu_baseline.simd_value): the same chain on fft_simd.c's vectorisable radix-4 Stockham transform */
long sgzo_spectrogram_range_simd(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                                 long f0, long f1, uint8_t *rgba_out);
int  sgzo_fft_forward_simd(sgzo_cf *buf, uint32_t N);    /* fft_simd.c; same definition as sgzo_fft_forward */
long sgzo_num_frames(size_t nsamples, uint32_t W, uint32_t hop);
/* frequency tracker, raw-FFT branch (SpectrumRendering.cpp:379-469): out = {peakOffset, peakFraction, peakFrequency, peakDBs, alpha, beta, gamma, phi} */
void sgzo_track_peak(const sgzo_spectrum_params *p, const sgzo_cf *source, uint32_t N, const float *mapped, double window_scale,
                     double mouse_fraction, double out[8]);
void sgzo_track_peak_lines(const sgzo_spectrum_params *p, const float *results, const float *mapped, const float *slope, uint32_t transform_size,
                           double mouse_fraction, double out[6]);
/* test hooks: a5 + a7 over F frames of given csp values [F][C][2P] (states start from zero); libm logf over an array */
long sgzo_decay_colour(const sgzo_spectrum_params *p, const sgzo_cf *csp_all, long F, uint8_t *rgba_out, sgzo_cf *line_out);
void sgzo_logf_array(const float *x, float *y, size_t n);

/* ---------------- the Spectrum view as a stream (spectrum_stream.c): block structure, quirks Q1 / Q2, the line-graph render path ---------------- */
/* SpectrumContent::DisplayMode (Source/Spectrum/SpectrumParameters.h): LineGraph = the render thread transforms the current history
 * once per video frame (SpectrumRendering.cpp:617-635); ColourSpectrum = the audio thread emits a frame every `hop` samples */
enum { SGZO_DISPLAY_LINE_GRAPH = 0, SGZO_DISPLAY_COLOUR_SPECTRUM = 1 };
typedef struct sgzo_spectrum_stream sgzo_spectrum_stream;
/* history: audioHistorySize in samples (0 = the window size, what Spectrum.cpp:472-477 asks for; larger values exercise quirk Q2) */
sgzo_spectrum_stream *sgzo_stream_create(const sgzo_spectrum_params *p, uint32_t display_mode, size_t history);
void   sgzo_stream_destroy(sgzo_spectrum_stream *s);
/* onStreamAudio -> AudioDispatcher::dispatch (SpectrumDSP.cpp:63-108) -> audioEntryPoint (TransformDSP.inl:1165-1211, block
 * prepareTransform :234-484) -> blendAndDispatchSpectrums; then the block enters the history.  rgba_out [max_frames][P][4]; line_out
 * [pair][graph][P] = lineGraphs[k].results after the callback; mapped_out [max_frames][pair][2P] = every frame's csp.  Returns the
 * number of frames the callback produced. */
long   sgzo_stream_audio(sgzo_spectrum_stream *s, const float *const *planar, size_t n, uint8_t *rgba_out, sgzo_cf *line_out,
                         sgzo_cf *mapped_out, size_t max_frames);
/* vectorGLRendering's LineGraph case (SpectrumRendering.cpp:617-635; whole-ring prepareTransform TransformDSP.inl:39-231):
 * results [pair][graph][P], mapped_out [pair][2P] (optional) */
int    sgzo_stream_render_lines(sgzo_spectrum_stream *s, sgzo_cf *results, sgzo_cf *mapped_out);
/* parity-chain twin: the filters of every pair advanced by one frame of GIVEN csp values [pair][2P] */
void   sgzo_stream_filters_given(sgzo_spectrum_stream *s, const sgzo_cf *csp_all, sgzo_cf *results, uint8_t *rgba_out);
void   sgzo_stream_history(const sgzo_spectrum_stream *s, uint32_t channel, size_t count, float *out);
size_t sgzo_stream_counter(const sgzo_spectrum_stream *s);           /* pairs[0].processedSamplesSinceLastFrame */

/* ---------------- Oscilloscope (a10..a12) ---------------- */
typedef struct sgzo_zero_crossing_state {   /* Source/Oscilloscope/StreamPreprocessing.h:315-349 */
    double   state;
    double   threshold;
    uint64_t steady_clock;
    uint64_t cross_origin;
    uint64_t count;
    int      armed;                          /* isPeakHolding */
} sgzo_zero_crossing_state;
/* returns number of triggers written (at most max_out); indices are absolute sample positions */
size_t sgzo_zero_crossing_process(sgzo_zero_crossing_state *st, uint32_t osc_mode, const float *a,
                                  const float *b, size_t n, uint64_t *out, size_t max_out);

typedef struct sgzo_scope_view {            /* the scalars drawWavePlot derives, OscilloscopeRendering.cpp:551-649 */
    double window_size;      /* state.effectiveWindowSize (samples)       */
    double left, right;      /* state.viewOffsets[Left/Right]             */
    double rendering_scale;  /* oglc->getRenderingScale()                 */
    uint32_t width;          /* getWidth() pixels                         */
    uint32_t _pad;
} sgzo_scope_view;
/* ring: history with newest sample at ring[len-1]; out_xy: (x,y) float pairs; returns point count */
size_t sgzo_scope_lanczos(const sgzo_scope_view *v, const float *ring, size_t len,
                          float *out_x, float *out_y, size_t max_points);
size_t sgzo_scope_num_points(const sgzo_scope_view *v);

/* drawWavePlot for one evaluator on front-buffer memory + cursor (trigger_mode 0 None / 4 ZeroCrossing; interpolation 2 Linear / 3 Lanczos) */
size_t sgzo_scope_wave_plot_ex2(const sgzo_scope_view *v, int trigger_mode, int interpolation, const float *memA, const float *memB,
                                int eval_mode, size_t size, size_t cursor, double cycle_samples, double sample_offset,
                                int64_t transport_position, uint32_t key, const uint32_t *colour_mem, float *xyz, uint32_t *rgba,
                                size_t max_points);
size_t sgzo_scope_wave_plot(const sgzo_scope_view *v, int trigger_mode, int interpolation, const float *memA, const float *memB,
                            int eval_mode, size_t size, size_t cursor, float *xyz, size_t max_points);

/* a12: peak envelope, OscilloscopeDSP.inl:713-886 / VectorscopeRendering.cpp:826-889 */
double sgzo_peak_filter(const float *const *ch, uint32_t nch, size_t n, uint32_t lanes,
                        double coeff_pow, double *env /*nch, in/out*/);

/* ---------------- Oscilloscope audio-thread state machine (a11, a12; scope_stream.c) ---------------- */
/* StreamState::audioEntryPoint for TriggeringMode None (0) / ZeroCrossing (4): trigger detection, TriggeringProcessor::processMutating
 * (window selection, back -> front swapBuffers), RMS envelope, ring writes.  env_mode: 0 none, 1 RMS, 2 peak decay. */
typedef struct sgzo_scope_stream sgzo_scope_stream;
sgzo_scope_stream *sgzo_scope_stream_create(uint32_t channels, double sample_rate, double window_size, int trigger_mode,
                                            double threshold, uint32_t osc_mode, double trigger_channel_1based,
                                            uint32_t env_mode, double envelope_window_s);
void   sgzo_scope_stream_destroy(sgzo_scope_stream *s);
void   sgzo_scope_stream_set_hysteresis(sgzo_scope_stream *s, double hysteresis);
void   sgzo_scope_stream_audio(sgzo_scope_stream *s, const float *const *planar, size_t n);
size_t sgzo_scope_stream_size(const sgzo_scope_stream *s);                          /* front buffer size = ceil(window + 1) */
size_t sgzo_scope_stream_front(const sgzo_scope_stream *s, uint32_t c, float *out); /* raw ring memory; returns the cursor */
double sgzo_scope_stream_envelope_gain(const sgzo_scope_stream *s);
void   sgzo_scope_stream_envelopes(const sgzo_scope_stream *s, float *out);
void   sgzo_scope_stream_state(const sgzo_scope_stream *s, uint64_t out[8]);
double sgzo_scope_stream_peak_filter(sgzo_scope_stream *s, uint32_t lanes, double coeff);   /* runPeakFilter, all channel modes */
void   sgzo_scope_stream_enable_colours(sgzo_scope_stream *s, const float band_colours[3][3], double frequency_colouring_blend,
                                        double colour_smoothing_ms, const uint8_t (*keys)[4]);
size_t sgzo_scope_stream_front_colours(const sgzo_scope_stream *s, uint32_t c, int aux, uint32_t *out);
size_t sgzo_scope_wave_plot_ex(const sgzo_scope_view *v, int trigger_mode, int interpolation, const float *memA, const float *memB,
                               int eval_mode, size_t size, size_t cursor, double cycle_samples, double sample_offset,
                               const uint32_t *colour_mem, float *xyz, uint32_t *rgba, size_t max_points);

/* ---------------- Oscilloscope spectral trigger + frequency colouring (scope_spectral.c) ---------------- */
typedef struct sgzo_bin_record { uint64_t index; double value, offset; } sgzo_bin_record;      /* Oscilloscope.h BinRecord */
typedef struct sgzo_spectral_state {                                                           /* triggerState + the median filter */
    sgzo_bin_record median[8];        /* medianTriggerFilter[i].record, value-initialised */
    uint64_t median_pos;              /* medianPos */
    sgzo_bin_record record;           /* triggerState.record */
    double fundamental, cycle_samples, sample_offset, phase;
} sgzo_spectral_state;
void sgzo_nth_element_by_index(sgzo_bin_record *v, int n, int nth);
void sgzo_scope_fundamental(sgzo_spectral_state *ts, const float *memA, const float *memB, int eval_mode, size_t size, size_t cursor,
                            double window_size, double sample_rate, double threshold, double hysteresis);
void sgzo_scope_fundamental_custom(sgzo_spectral_state *ts, double custom_frequency, double sample_rate);
void sgzo_scope_trigger_offset(sgzo_spectral_state *ts, const float *memA, const float *memB, int eval_mode, size_t size, size_t cursor,
                               double window_size, double sample_rate, double phase_offset_degrees);
typedef struct sgzo_lr_coeffs { float lp1[5], hp1[5], lp2[5], hp2[5]; } sgzo_lr_coeffs;      /* b0 b1 b2 a1 a2 per section type */
typedef struct sgzo_lr_state { float z[8][2]; } sgzo_lr_state;                               /* lp1 a,b  hp1 a,b  lp2 a,b  hp2 a,b */
void  sgzo_lr_design(double low_hz, double high_hz, double sample_rate, sgzo_lr_coeffs *out);
void  sgzo_lr_process(sgzo_lr_state *st, const sgzo_lr_coeffs *k, float x, float bands[3]);
float sgzo_colour_smooth_pole(double milliseconds, double sample_rate);
void  sgzo_colour_filter_states(const float bands[3], float states[3], float pole);
void  sgzo_colour_accumulate(const float state[3], const float colours[3][3], const uint8_t key[4], float blend, uint8_t out[4]);
void  sgzo_colour_lerp_f64(const uint8_t a[4], const uint8_t b[4], double t, uint8_t out[4]);

/* ---------------- Vectorscope (a13, a14) ---------------- */
void sgzo_vector_polar(const float *L, const float *R, size_t n, int fade, float *xyz /*n*3*/);
/* drawPolarPlot over a two-section view of the history ring (memory + write cursor), fade ramp and colours included */
void sgzo_vector_polar_view(const float *memL, const float *memR, size_t size, size_t cursor, uint32_t lanes, int fade_history,
                            const float colour[3], float *xyz /*size*3*/, float *rgb_out /*size*3 or NULL*/);
typedef struct sgzo_vector_filters {          /* Source/Vectorscope/Vectorscope.h filters */
    float env[2];
    float balance[2][2];
    float phase[2];
} sgzo_vector_filters;
void sgzo_vector_audio_processing(sgzo_vector_filters *f, const float *L, const float *R, size_t n,
                                  uint32_t lanes, float envelope_coeff, float stereo_coeff,
                                  float second_speed, int env_mode /*0 none,1 rms*/, float *gain_out);

/* ---------------- RSNT: the resonator algorithm (resonator.c; cpl::dsp::CComplexResonator restated, UNVERIFIED vs cpl) ---------------- */
int  sgzo_window_cosine_terms(uint32_t window_type, double a[SGZO_RES_MAX_TERMS]);
void sgzo_resonator_map(const sgzo_spectrum_params *p, const float *mapped, sgzo_cf *coeff /*[V][P]*/, float *gain /*[P]*/,
                        float *weights /*[V]*/, int *vectors);
void sgzo_resonate_real(const sgzo_cf *coeff, uint32_t P, int V, sgzo_cf *state /*[signals][V][P]*/, const float *const *work,
                        int signals, size_t n);
int  sgzo_resonator_dispatch(uint32_t mode, const float *L, const float *R, size_t n, float *work0, float *work1);
void sgzo_resonator_windowed_state(const sgzo_spectrum_params *p, const sgzo_cf *state, const float *gain, const float *weights,
                                   int V, int signals, sgzo_cf *csp /*[2P]*/);
long sgzo_resonator_num_frames(size_t nsamples, uint32_t hop);
long sgzo_resonator_spectrogram(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                                uint8_t *rgba_out, sgzo_cf *line_out, sgzo_cf *mapped_out);
long sgzo_resonator_spectrogram_scaled(const sgzo_spectrum_params *p, const float *const *planar, size_t nsamples,
                                       uint8_t *rgba_out, sgzo_cf *line_out, sgzo_cf *mapped_out, float *scale_out /*[F][C][2][P]*/);

#ifdef __cplusplus
}
#endif
#endif
