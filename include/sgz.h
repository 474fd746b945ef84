/*
 * sgz.h -- C ABI of libsgz.so, the MI355X (gfx950) DSP back end for Signalizer's visualiser hot path.
 *
 * Drop-in boundary.  Every entry point names the reference interface it
 * replaces (paths relative to jthorborg/signalizer v0.4.3).  Plain pointers and sizes only: no C++
 * types, no torch types, no exceptions cross this boundary.  All functions return an sgz_status;
 * sgz_last_error() gives the text of the last failure on the calling thread.
 *
 * Pointer spaces: parameters named `d_*` are DEVICE (HBM) pointers, everything else is host memory.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 */
#ifndef SGZ_H
#define SGZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: sgz_spectrum_config grew algorithm / free_q, sgz_scope_config custom_trigger / custom_trigger_frequency (round 3);
 * 4: sgz_spectrum_config grew display_mode (ZERO = the line graph, as in the reference's enum), sgz_spectrum_render_lines,
 *    sgz_spectrum_set_option (round 4);
 * 5: sgz_scope_set_option / sgz_vector_set_option (SGZ_RT_OPT_DEFER_SUBMIT, SGZ_RT_OPT_PARK_PUSHES), sgz_spectrum_track_peak_lines, plan option
 *    SGZ_OPT_WIDE_GROUPS; the Oscilloscope / Vectorscope readers flush the host FIFO as well; SGZ_OPT_RESONATOR_SLAB bounds the sharded RSNT render (round 5; its own option SGZ_OPT_RESONATOR_SHARD_BOUND from round 6);
 * a binding compares sgz_abi_version() with the header it was compiled against */
#define SGZ_ABI_VERSION 5

typedef enum sgz_status {
    SGZ_OK = 0,
    SGZ_EMPTY = 1,          /* nothing to pop (frameQueue empty)                              */
    SGZ_SKIPPED_FRAME = 2,  /* prepareTransform returned false, TransformDSP.inl:45-46          */
    SGZ_BUSY = 3,           /* a real-time push found the GPU several blocks behind (or a reconfiguration in progress): the block
                               was NOT taken -- push never waits                                   */
    SGZ_EINVAL = -1,
    SGZ_EHIP = -2,          /* a HIP runtime call failed / no gfx950 device                     */
    SGZ_ENOMEM = -3,
    SGZ_EUNSUPPORTED = -4   /* e.g. line results from a folded carry in SpectrumChannels::Phase     */
} sgz_status;

/* SpectrumChannels, Source/Common/CommonSignalizer.h:495-539 */
enum { SGZ_CH_LEFT = 0, SGZ_CH_RIGHT, SGZ_CH_MERGE, SGZ_CH_SIDE, SGZ_CH_PHASE, SGZ_CH_SEPARATE,
       SGZ_CH_MIDSIDE, SGZ_CH_COMPLEX };
/* SpectrumContent::TransformAlgorithm, Source/Spectrum/SpectrumParameters.h:66-69.  RSNT ("Resonator"): a bank of complex
 * one-pole resonators, one per axis point, advanced by every sample (TransformPair::resonatingDispatch, TransformDSP.inl:1213-1295);
 * a frame is the windowed resonator state every `hop` samples (audioEntryPoint :1172-1201, mapToLinearSpace :1103-1133).  The
 * resonator itself is cpl::dsp::CComplexResonator (absent submodule): restated from its published mathematics, see resonator.hip.
 * EXPERIMENTAL as a drop-in: Signalizer's own part of this algorithm (dispatch, cadence, Phase post-processing, everything behind the
 * frame) follows the reference line by line, but the bank's constants -- bandwidth = spacing to the next axis point, the Q bound by the
 * window size, pole radius exp(-pi B / fs), gain 1 - r, V = 2 K - 1 detuned vectors with the cosine-sum weights a_m / 2, the ignored
 * vectorLength argument of mapSystemHz -- are this build's reading of what such a bank must be, checked against mathematics (an
 * exponentially windowed DFT) and against nothing of cpl's.  A maintainer with the cpl sources should compare
 * sgz_plan_get_resonator() with CComplexResonator::Constant after mapSystemHz (and the vector count with
 * cpl::dsp::windowCoefficients(window).second) before shipping it; the FFT algorithm carries no such caveat beyond "parity unpinned". */
enum { SGZ_ALGO_FFT = 0, SGZ_ALGO_RSNT = 1 };
/* SpectrumContent::DisplayMode, Source/Spectrum/SpectrumParameters.h:60-64 (constant.displayMode, Spectrum.cpp:439).  Only the real-time
 * handle reads it: LINE_GRAPH -- the audio thread transforms nothing (TransformDSP.inl:1167; RSNT: it keeps the resonators running,
 * :1206-1209) and the render thread transforms the current history once per video frame (sgz_spectrum_render_lines =
 * SpectrumRendering.cpp:617-635); COLOUR_SPECTRUM -- a frame every `hop` samples on the audio thread, columns through the frame queue.
 * Plans and the offline / stage entry points ignore it (they are the colour spectrum's chain). */
enum { SGZ_DISPLAY_LINE_GRAPH = 0, SGZ_DISPLAY_COLOUR_SPECTRUM = 1 };
/* SpectrumContent::BinInterpolation */
enum { SGZ_INTERP_NONE = 0, SGZ_INTERP_LINEAR, SGZ_INTERP_LANCZOS };
/* SpectrumContent::ViewScaling */
enum { SGZ_VIEW_LINEAR = 0, SGZ_VIEW_LOG };
/* window shapes / symmetry (stand-in for cpl::dsp::WindowTypes, which lives in the missing cpl) */
enum { SGZ_WIN_RECT = 0, SGZ_WIN_HANN, SGZ_WIN_HAMMING, SGZ_WIN_FLATTOP, SGZ_WIN_BLACKMAN,
       SGZ_WIN_EXACT_BLACKMAN, SGZ_WIN_NUTTALL, SGZ_WIN_BLACKMAN_NUTTALL, SGZ_WIN_BLACKMAN_HARRIS,
       SGZ_WIN_TRIANGULAR, SGZ_WIN_WELCH, SGZ_WIN_GAUSSIAN, SGZ_WIN_KAISER, SGZ_WIN_END };
enum { SGZ_WIN_SYMMETRIC = 0, SGZ_WIN_PERIODIC };
/* OscChannels, Source/Common/CommonSignalizer.h:458-493 */
enum { SGZ_OSC_LEFT = 0, SGZ_OSC_RIGHT, SGZ_OSC_MID, SGZ_OSC_SIDE, SGZ_OSC_SEPARATE, SGZ_OSC_MIDSIDE };

/* OscilloscopeContent::TriggeringMode, Source/Oscilloscope/OscilloscopeParameters.h:50-58 */
enum { SGZ_TRIG_NONE = 0, SGZ_TRIG_SPECTRAL, SGZ_TRIG_WINDOW, SGZ_TRIG_ENVELOPE_HOLD, SGZ_TRIG_ZERO_CROSSING };
/* EnvelopeModes / SubSampleInterpolation, Source/Common/CommonSignalizer.h:72-85 */
enum { SGZ_ENV_NONE = 0, SGZ_ENV_RMS, SGZ_ENV_PEAK_DECAY };
enum { SGZ_SUBSAMPLE_NONE = 0, SGZ_SUBSAMPLE_RECTANGULAR, SGZ_SUBSAMPLE_LINEAR, SGZ_SUBSAMPLE_LANCZOS };

#define SGZ_NUM_SPEC_COLOURS 5   /* SpectrumContent::numSpectrumColours */
#define SGZ_NUM_GRAPHS 2         /* SpectrumContent::LineGraphs::LineEnd (LineMain, LineSecond) */

/* POD mirror of Signalizer::TransformConstant<float> inputs
 * (Source/Spectrum/TransformConstant.h:190-239, filled by Spectrum::handleFlagUpdates,
 *  Source/Spectrum/Spectrum.cpp:351-616). */
typedef struct sgz_spectrum_config {
    float    sample_rate;
    uint32_t window_size;        /* W; transform size N = max(32, nextPow2(W)), TransformConstant.h:84 */
    uint32_t hop;                /* sampleBufferSize, SpectrumDSP.cpp:51-54                           */
    uint32_t axis_points;        /* P, Spectrum.cpp:445                                               */
    uint32_t channel_mode;       /* SGZ_CH_*                                                          */
    uint32_t bin_interp;         /* SGZ_INTERP_*                                                      */
    uint32_t view_scaling;       /* SGZ_VIEW_*                                                        */
    uint32_t window_type;        /* SGZ_WIN_*                                                         */
    uint32_t window_symmetry;
    uint32_t num_pairs;          /* stereo pairs (numChannels/2), SpectrumDSP.cpp:65-72               */
    double   window_alpha, window_beta;
    double   view_left, view_right;
    double   min_log_freq;
    double   low_db, high_db, clip_db;
    double   slope_a, slope_b;
    float    pole[SGZ_NUM_GRAPHS];                    /* constant.filter[k].pole                      */
    uint8_t  colours[SGZ_NUM_SPEC_COLOURS + 1][3];    /* [0] background, [1..5] gradient stops (RGB8) */
    uint8_t  _pad[2];
    double   ratios[SGZ_NUM_SPEC_COLOURS];            /* content->specRatios (normalised values)      */
    uint32_t algorithm;          /* SGZ_ALGO_*: constant.algo, Spectrum.cpp:367                                  */
    uint32_t free_q;             /* RSNT: content->freeQ (Spectrum.cpp:593): bandwidths not bounded by the window */
    uint32_t display_mode;       /* SGZ_DISPLAY_*: constant.displayMode (real-time handle only)                   */
    uint32_t _reserved;
} sgz_spectrum_config;

typedef struct sgz_timing {         /* filled by the batch entry points when non-NULL */
    double h2d_ms, kernel_ms, d2h_ms;
    uint64_t frames;
} sgz_timing;

const char *sgz_last_error(void);
int         sgz_abi_version(void);
/* number of visible gfx950 devices (0 when no GPU: every compute entry point then fails with SGZ_EHIP;
 * there is NO CPU fallback in this library). */
int         sgz_device_count(void);
sgz_status  sgz_set_device(int device);

/* ------------------------------------------------------------------------------------------------
 * Spectrum "constant block": replaces TransformConstant<float> + its (re)build in
 * Spectrum::handleFlagUpdates (Spectrum.cpp:489 setStorage, :573 remapFrequencies,
 * :580 generateSlopeMap, :586 regenerateWindowKernel, :226-246 colour ratios).
 * Host-side tables are built on the CPU in fp64 with the reference's expression order and uploaded
 * once; the query functions below expose them for parity tests (they need no GPU).
 *
 * Threading: a plan belongs to the device that was current at its first compute call, and owns the
 * per-launch scratch of its kernels -- like the reference's TransformConstant + TransformPair, which one
 * audio thread uses at a time.  One host thread / stream per plan at a time; concurrent renders use one plan
 * each (the tables are small).  The library itself keeps no process-wide device state: the stage calls'
 * scratch is stream-ordered, the real-time handles own theirs.
 */
typedef struct sgz_plan sgz_plan;
sgz_status sgz_plan_create(const sgz_spectrum_config *cfg, sgz_plan **out);   /* host tables only, no GPU needed */
void       sgz_plan_destroy(sgz_plan *plan);
sgz_status sgz_plan_upload(sgz_plan *plan);                                   /* device tables; needs a GPU */
uint32_t   sgz_plan_transform_size(const sgz_plan *plan);
double     sgz_plan_window_scale(const sgz_plan *plan);                        /* windowKernelScale */
uint32_t   sgz_plan_break_pixel(const sgz_plan *plan);                         /* first max-of-bins pixel */
/* which K_A implementation the transform size and channel mode select: SGZ_PATH_FUSED: N = R^3
 * (4096, 32768) in one workgroup; SGZ_PATH_HALVES: N = 2 R^3 (8192, 65536) as two half-frame workgroups + a map kernel;
 * SGZ_PATH_GENERIC: HBM-resident passes (every other size, and Phase at any size).  Bit SGZ_PATH_SIDE_MAP: the halves /
 * generic path can use the LDS-staged per-side map kernel (the view's records stay inside the staged csf range). */
#define SGZ_PATH_GENERIC  0u
#define SGZ_PATH_FUSED    1u
#define SGZ_PATH_HALVES   2u
#define SGZ_PATH_SIDE_MAP 4u
#define SGZ_PATH_CHANNEL_SPLIT 8u   /* N = 16384 / 32768 / 65536, W == N, even hop: the real-input kernel (spectrum_real.hip) -- Separate: one
                                       workgroup per (frame, pair, channel); Left / Right / Merge / Side: one per (frame, pair) on the mixed
                                       signal -- takes the place of the kernels above, whatever the row layout
                                       (at N = 32768 in Separate mode: for launches of up to 1024 tasks) */
uint32_t   sgz_plan_path(const sgz_plan *plan);
/* frames an offline render / stage call produces from `nsamples` samples per channel: FFT: sgz_num_frames(nsamples, W, hop);
 * RSNT: nsamples / hop (a frame after every hop samples, no window history) */
uint64_t   sgz_plan_num_frames(const sgz_plan *plan, size_t nsamples);
/* RSNT plans: the resonator bank CComplexResonator::Constant::mapSystemHz builds (TransformConstant.h:120-123) -- `vectors` detuned
 * resonators per axis point: coeff [vectors][P] (re, im) pole positions, gain [P], weights [vectors] of the frequency-domain window.
 * Any output may be NULL; returns SGZ_EINVAL on an FFT plan. */
sgz_status sgz_plan_get_resonator(const sgz_plan *plan, uint32_t *vectors, float *coeff, float *gain, float *weights);
/* RSNT plans carry the resonator state between calls (the reference's TransformPair::cresonator).  sgz_stage_mapped and renders without
 * a carried decay state (d_state == NULL) start from rest by themselves; a render WITH d_state continues the stream, resonators
 * included.  This call puts them to rest explicitly (TransformPair.h:183 resetState); no-op on FFT plans. */
sgz_status sgz_plan_reset_resonator(sgz_plan *plan, void *stream);
/* Per-plan switches (all default to what is fastest; the parity tests and measurements flip them):
 *   SGZ_OPT_CHANNEL_SPLIT  1 (default): eligible plans run the real-input channel-split kernels (SGZ_PATH_CHANNEL_SPLIT); 0: never
 *   SGZ_OPT_FUSED_COLOUR   1 (default): an image-only K_B of one pair runs as the single fused launch, a workgroup per 4 pixels; 8 / 16:
 *                          the same with 8 / 16 pixels per workgroup (fewer, longer workgroups: slower on an idle device, but less of
 *                          the chip is taken from kernels that run beside it -- what sgz_render_queue lanes use); 0: scan + emit launches
 *   SGZ_OPT_FETCH_WINDOW   0 (default): Hann / Hamming periodic windows at W == N are evaluated inside K_A; 1: fetched from the table
 * Call between sgz_plan_create and the first compute call on the plan (same threading rule as every other plan call). */
#define SGZ_OPT_CHANNEL_SPLIT 1u
#define SGZ_OPT_FUSED_COLOUR  2u
#define SGZ_OPT_FETCH_WINDOW  3u
#define SGZ_OPT_MATRIX_RESONATOR 4u /* RSNT launches of several frames at a hop that is a multiple of 1024: every frame starts from rest as block
                                      sums on the matrix cores and the frames are chained afterwards (a one-frame launch -- the real-time case --
                                      is always the reference's recurrence sample by sample).  2 (default): the fp32 matrix cores
                                      (resonateMfmaKernel, exact fp32 multiply-add chains); 1: the bf16 matrix cores, every fp32 sample and
                                      weight as the exact sum of three bf16 parts, six part products (fp32-equivalent accuracy, twice as fast:
                                      resonator.hip resonateMfmaBf16Kernel) -- OPT-IN since round 6: on the MI355X boxes this was measured on, FFT
                                      kernels that run on the device at the same time (another stream, another process; this library's and
                                      rocFFT's alike) came back with a wrong cache line's worth of values in 2 of 100 000 launches while this
                                      kernel ran in its round-4 form, and in every second launch with other instruction orders of the same
                                      kernel; the form shipped now (four idle cycles behind every matrix instruction, the middle of a range
                                      that was clean in 350 000 launches) showed none -- a measurement, not a guarantee: choose it when nothing
                                      else shares the device (NOTES.md, "A matrix-core kernel that disturbs its neighbours"); 0: the
                                      vector-ALU block form everywhere (frame 0 of a launch then continues the carried state sample by sample) */
#define SGZ_OPT_RESONATOR_SLAB 5u   /* RSNT: frames per slab of a long render (the per-frame resonator states between the kernels are held for one
                                      slab at a time; 0, the default: as many frames as fit 256 MiB).  A slab continues the state the one
                                      before it left.  It plays no part in the sharded render (SGZ_OPT_RESONATOR_SHARD_BOUND) */
#define SGZ_OPT_WIDE_GROUPS 6u       /* N = 32768 channel-split plans (pairs): 0 (default): one 512-thread workgroup per (frame, pair, channel), 32 values
                                      per thread (spectrum_real.hip); 1: 1024 threads of sixteen values (spectrum_real16.hip: 8 waves per SIMD; measured
                                      7-12 % slower on MI355X -- NOTES.md round 5 -- and kept as a tested alternative) */
#define SGZ_OPT_RESONATOR_SHARD_BOUND 7u /* RSNT, sgz_spectrogram_render_sharded only: that render holds the per-frame resonator states of a rank's WHOLE
                                      chunk between its two halves (from rest ... carry + windows) and cannot cut it, so this is a bound, in frames
                                      (0, the default: 8 GiB worth): a rank's chunk above it is refused with SGZ_EUNSUPPORTED on every rank before
                                      anything is allocated or exchanged (round 6; rounds 4-5 read SGZ_OPT_RESONATOR_SLAB here) */
#define SGZ_OPT_PIPELINED 8u          /* 1: other launches run beside this plan's (a lane of an sgz_render_queue sets it): K_A without the tunings of a launch
                                      that has the chip to itself (the second generation's delayed start, the wave priorities).  Same results. */
sgz_status sgz_plan_set_option(sgz_plan *plan, uint32_t option, uint32_t value);
/* The pixels whose filter taps or arg-max run reach a csf entry the reference leaves complex -- Complex: csf[0] = Z[0]/2
 * (TransformDSP.inl:993); Left / Right / Merge / Side: csf[N/2 .. N-1] (:553-560), reached by windows that wrap below bin 0 or
 * sit at Nyquist -- redone as complex sums after the magnitude-only mapping.  Returns their count; writes at most `cap`. */
uint32_t   sgz_plan_dc_pixels(const sgz_plan *plan, uint32_t *out, uint32_t cap);
sgz_status sgz_plan_get_window(const sgz_plan *plan, float *out /*N*/);
sgz_status sgz_plan_get_mapped_frequencies(const sgz_plan *plan, float *out /*P*/);
sgz_status sgz_plan_get_slope_map(const sgz_plan *plan, float *out /*P*/);
sgz_status sgz_plan_get_colour_ratios(const sgz_plan *plan, float *out /*6*/);
sgz_status sgz_plan_get_colour_table(const sgz_plan *plan, uint32_t pair, float *out /*6*3*/);
/* juce::Colour::withRotatedHue restated (juce_Colour.cpp:33-107,:331-336) */
void       sgz_rotate_hue_rgb8(const uint8_t rgb[3], float amount, uint8_t out[3]);
long       sgz_num_frames(size_t nsamples, uint32_t window_size, uint32_t hop);

/* ------------------------------------------------------------------------------------------------
 * Offline / batch spectrogram: TransformPair::audioEntryPoint (TransformDSP.inl:1165-1211) +
 * AudioDispatcher::blendAndDispatchSpectrums (SpectrumDSP.cpp:111-206) over a whole buffer with ideal
 * STFT framing (frame f = samples [f*hop, f*hop+W)).
 *
 * d_planar:   DEVICE fp32, channel c at d_planar + c*channel_stride, 2*num_pairs channels, nsamples each.
 * d_rgba:     DEVICE RGBA8 [frames][P][4]        (the columns oglImage.updateSingleColumn receives)
 * d_lines:    optional DEVICE float2 [frames][pairs][graphs][P] (lineGraphs[k].results, TransformPair.h:63-94)
 * d_state:    optional DEVICE float2 [pairs][graphs][P] peak-decay state, read as carry-in and
 *             updated to the state after the last frame (lineGraphs[k].states); NULL = start from zero.
 * Asynchronous on `stream`; the caller synchronises.  Fewer samples than one window: nothing is rendered and the call
 * returns SGZ_SKIPPED_FRAME (prepareTransform returns false, TransformDSP.inl:45-46); sgz_spectrogram_render likewise.
 */
sgz_status sgz_spectrogram_render_device(sgz_plan *plan, const float *d_planar, size_t channel_stride,
                                         size_t nsamples, uint8_t *d_rgba, float *d_lines,
                                         float *d_state, void *stream);
/* ---- render queue: a job of MANY buffers, several renders in flight (round 6) ------------------------------------------------------
 * One render of BASELINE configs[1] is 696 workgroups on 512 slots: two full generations and a third that leaves two thirds of the
 * chip idle, then K_B's dependent chain.  A job of many buffers (a folder of files, the channel pairs of a session rendered to separate
 * images, a long buffer cut into independent images) need not pay that per buffer: the queue owns `depth` lanes -- a plan (a plan owns
 * its scratch) and a stream each -- and submits round-robin, so that one buffer's K_B and the partly filled last generation of its K_A
 * run beside the next buffers' K_A.  Measured (tools/pipeline_depth.py, input rotated past the Infinity Cache): 33.3 us per render one
 * after the other, 25.3-26.1 us at depth >= 3: +27 ... 31 % frames/s.  The reference has no counterpart (it transforms one frame per
 * audio callback, TransformDSP.inl:1165-1211); results per buffer are those of sgz_spectrogram_render_device(plan, ..., d_lines = NULL,
 * d_state = NULL, ...) byte for byte (every buffer is a job of its own: decay states start from zero).
 *   submit   never waits for the GPU (a lane that is still busy simply queues the work behind its previous render).  The samples must be
 *            complete -- and d_rgba free of pending work of other streams (a fill, an earlier reader) -- when the render starts:
 *            `after_stream` (may be NULL) is the caller's stream that produced / last touched them; the lane waits for what that stream
 *            holds at the time of the call.  *ticket (optional) numbers the submission, from 1.
 *   wait     host wait until submission `ticket` has finished (0: everything submitted so far)
 *   join     makes `stream` (the caller's) wait for everything submitted so far, without a host wait
 * One thread at a time per queue.  depth 1 .. 16 (3 or more reaches the plateau). */
typedef struct sgz_render_queue sgz_render_queue;
sgz_status sgz_render_queue_create(const sgz_spectrum_config *cfg, uint32_t depth, sgz_render_queue **out);
void       sgz_render_queue_destroy(sgz_render_queue *q);
sgz_status sgz_render_queue_submit(sgz_render_queue *q, const float *d_planar, size_t channel_stride, size_t nsamples,
                                   uint8_t *d_rgba /*[frames][P][4]*/, void *after_stream, uint64_t *ticket);
sgz_status sgz_render_queue_wait(sgz_render_queue *q, uint64_t ticket);
sgz_status sgz_render_queue_join(sgz_render_queue *q, void *stream);
/* Lanes whose streams run side by side.  The runtime maps streams onto a few hardware queues (4 by default; GPU_MAX_HW_QUEUES) and two
 * lanes on one queue run one after the other, so sgz_render_queue_create picks its streams by measurement (a 100 us spin kernel per
 * stream); with depth above the number of hardware queues the remaining lanes share.  Diagnostic. */
uint32_t   sgz_render_queue_distinct_lanes(const sgz_render_queue *q);
/* the lane plans' option (sgz_plan_set_option on every lane; before the first submit) */
sgz_status sgz_render_queue_set_option(sgz_render_queue *q, uint32_t option, uint32_t value);

/* Host buffers in, host buffers out (H2D, render, D2H on a stream of the plan's own); `planar` are the reference's planar channel
 * pointers (AudioStream::Listener::onStreamAudio's float** buffer, Spectrum.h:370).  The plan keeps the device copies between calls:
 * repeated renders of the same shape allocate nothing and rebuild no tables.  sgz_spectrogram_render is the one-shot form (plan
 * built and destroyed inside the call: the fp64 constant block costs more than the render itself). */
sgz_status sgz_spectrogram_render_host(sgz_plan *plan, const float *const *planar, uint32_t num_channels, size_t nsamples,
                                       uint8_t *rgba_out, float *lines_out, sgz_timing *timing);
sgz_status sgz_spectrogram_render(const sgz_spectrum_config *cfg, const float *const *planar,
                                  uint32_t num_channels, size_t nsamples, uint8_t *rgba_out,
                                  float *lines_out, sgz_timing *timing);

/* Device memory for the display hand-off: `bytes` rounded up to whole 2 MiB blocks (*allocated), exported as a dma-buf
 * file descriptor (dmabuf_fd may be NULL: plain allocation) that the GL / Vulkan context of the display GPU -- an MI355X has no graphics
 * engine of its own -- or any other process imports (EXT_memory_object_fd, EGL_EXT_image_dma_buf_import, hipImportExternalMemory with
 * hipExternalMemoryHandleTypeOpaqueFd and size = *allocated).  Whole blocks because a dma-buf is a whole buffer object and the runtime
 * packs smaller allocations into shared ones: the importer sees the memory from offset 0 (tests/test_gpu_realtime.py imports the fd in
 * a second process and compares the texels).  The caller closes the fd and frees the memory with sgz_export_free. */
sgz_status sgz_export_alloc(size_t bytes, void **d_ptr, size_t *allocated, int *dmabuf_fd);
void       sgz_export_free(void *d_ptr);

/* Stage entry points (parity tests call these through the ABI; all DEVICE pointers, async on stream):
 *  bins:   per (frame,pair) the post-split magnitude array csf[0..N] of mapToLinearSpace
 *          (TransformDSP.inl:858-869 for Separate/MidSide; :553-560 mono modes) as float [N+1];
 *  mapped: csp magnitudes after pixel mapping (TransformDSP.inl:871-985), float [frames][pairs][2][P];
 *  decay+colour from given mapped magnitudes (TransformDSP.inl:1299-1435 + SpectrumDSP.cpp:111-206).
 * SGZ_CH_PHASE (TransformDSP.inl:643-853, :1393-1432): the bins stay complex -- `bins` is float2 [N+1] (re, im) after
 * separateTransformsIPL and the DC / Nyquist fix-ups; the two planes of `mapped` are wsp[2x] (magnitude) and wsp[2x+1]
 * (phase cancellation); state and line results hold (magnitude, phase) where the other modes hold (left, right). */
sgz_status sgz_stage_bins(sgz_plan *plan, const float *d_planar, size_t channel_stride, size_t nsamples,
                          float *d_bins /*[frames][pairs][N+1]*/, void *stream);
sgz_status sgz_stage_mapped(sgz_plan *plan, const float *d_planar, size_t channel_stride, size_t nsamples,
                            float *d_mapped /*[frames][pairs][2][P]*/, void *stream);
sgz_status sgz_stage_map_from_bins(sgz_plan *plan, const float *d_bins, size_t frames,
                                   float *d_mapped, void *stream);
/* K_A's dominant launch ALONE, for timing it with events on `stream` (bench.py's roofline line): on a channel-split plan the pixels
 * that need both channels (the top pixels csf[N/2] can win, taps that reach over bin 0) hold the values of the channel's own bins
 * only -- sgz_stage_mapped / the render calls complete them (a small follow-up launch, or K_B's fused kernel as it reads them);
 * on every other plan this is sgz_stage_mapped. */
sgz_status sgz_stage_mapped_dominant(sgz_plan *plan, const float *d_planar, size_t channel_stride, size_t nsamples,
                                     float *d_mapped /*[frames][pairs][2][P]*/, void *stream);
/* d_rgba and d_lines may both be NULL: a state-only pass that just advances d_state over `frames` frames (what the
 * multi-GPU carry exchange below needs from every rank before the real pass). */
sgz_status sgz_stage_decay_colour(sgz_plan *plan, const float *d_mapped, size_t frames,
                                  uint8_t *d_rgba, float *d_lines, float *d_state, void *stream);

/* Frequency tracker, the raw-FFT branch of Spectrum::drawFrequencyTracking (Source/Spectrum/SpectrumRendering.cpp:379-469; SURVEY 8(f)
 * #4): nearest peak of the raw transform around the mouse position (fraction of the frequency axis, :267 / :292), walk along a rising
 * edge at the range boundary, parabolic fit in the dB domain.  peak_dbs is the value before the slope correction (:453: the caller
 * adds 20 log10(slopeMap[x])).  d_bins: DEVICE csf magnitudes [N + 1] of one (frame, pair) as sgz_stage_bins writes them.
 * Magnitude modes except Complex (:301).  The call waits for its result. */
typedef struct sgz_peak {
    double peak_offset;          /* bin of the peak                         */
    double peak_fraction;        /* 2 (bin + phi) / N                       */
    double peak_frequency;       /* Hz                                      */
    double peak_dbs;
    double alpha, beta, gamma;   /* 20 log10 of the three bins around it    */
    double phi;                  /* fractional bin offset of the parabola   */
} sgz_peak;
sgz_status sgz_stage_track_peak(sgz_plan *plan, const float *d_bins, double mouse_fraction, sgz_peak *out, void *stream);
/* The tracker's OTHER branch (SpectrumRendering.cpp:300-377): in Complex mode, for the RSNT algorithm and for the LineMain / LineSecond
 * graphs the reference looks for the peak in lineGraphs[graph].getResults(axisPoints) -- the displayed line -- instead of the raw bins:
 * first largest leftMagnitude within +-3 % of the axis around the mouse position, the walk along a still rising edge at a boundary of
 * that range, then peakFrequency = mapFrequency(peak), peakDeviance from the neighbouring axis points (for the FFT algorithm with a
 * non-Lanczos bin interpolation at least half a bin in axis points, :355-358), peakFractionY = the line's value there and its dB
 * value on the view's dB range.  `results`: HOST float2 [P] as sgz_spectrum_line_results / sgz_spectrum_render_lines deliver them --
 * the reference reads the same host-resident display results on its GUI thread; this is host arithmetic on a few thousand floats,
 * nothing is launched.  (peak_dbs = low_db + y (high_db - low_db): cpl::Math::UnityScale::linear, absent, by its name.) */
typedef struct sgz_line_peak {
    double peak_offset;          /* axis point of the peak (peakX)               */
    double peak_frequency;       /* mappedFrequencies[peak], Hz                  */
    double peak_deviance;        /* Hz                                           */
    double peak_fraction_y;      /* results[peak].leftMagnitude                  */
    double peak_dbs;
    double peak_slope;           /* slopeMap[peak]                               */
} sgz_line_peak;
sgz_status sgz_track_peak_lines(const sgz_plan *plan, const float *results /*HOST float2 [P]*/, double mouse_fraction, sgz_line_peak *out);

/* K_B in two steps, for the multi-GPU carry exchange (collective A2).  scan: the chunk scans of `frames` frames from a
 * ZERO carry-in; writes that zero-carry end state (what a rank publishes) to d_end_state [pairs][graphs][P][2] and keeps the chunk
 * aggregates inside the plan.  emit: folds the true carry-in d_carry (NULL = zero) into the kept aggregates -- one pass over the
 * aggregates, no second scan of the magnitudes -- and renders; d_state_out (optional) receives the state after the last frame.
 * Result == sgz_stage_decay_colour(..., d_state = carry) bit for bit.
 * SGZ_CH_PHASE: the magnitude half of the state -- a peak decay, and all the image is coloured from (SpectrumDSP.cpp:123) -- folds
 * the same way; the cancellation smoother (:1409-1412) is a linear fp32 recurrence without an exact fold, so emit renders the image
 * only (d_lines / d_state_out: SGZ_EUNSUPPORTED) and the phase halves of the scan's end state are not meaningful. */
sgz_status sgz_stage_decay_scan(sgz_plan *plan, const float *d_mapped, size_t frames, float *d_end_state, void *stream);
sgz_status sgz_stage_decay_emit(sgz_plan *plan, const float *d_mapped, size_t frames, const float *d_carry, uint8_t *d_rgba,
                                float *d_lines, float *d_state_out, void *stream);

/* std::log(float) as the dB map evaluates it (TransformDSP.inl:1345): glibc's logf algorithm, bit-identical to libm over every
 * positive finite float (tests/test_gpu_spectrum.py checks all 2^31 of them).  d_x > 0; DEVICE pointers. */
sgz_status sgz_stage_logf(const float *d_x, float *d_y, size_t n, void *stream);
/* the last step of every K_A pixel, magnitude = sqrt(re*re + im*im) with im == 0 (TransformDSP.inl:1331): evaluated as |x| where the
 * square is a normal float (the two are bit-identical there), as the correctly rounded root elsewhere */
sgz_status sgz_stage_finish_pixel(const float *d_x, float *d_y, size_t n, void *stream);

/* Multi-GPU time-chunk sharding (collective A2).  Rank q renders its frames with a zero
 * carry-in and publishes its end state A_q (the d_state output above).  Because fl(x*pole) is monotone,
 * the true state entering rank r is  fold_{q<r} carry = max(A_q, decay^{frames_q}(carry))  evaluated with
 * sequential fp32 multiplies -- bit-identical to the reference's sequential recurrence
 * (TransformDSP.inl:1336-1341) run over the whole stream on one device.
 * d_aggs: DEVICE float [world][pairs][graphs][P][2] (an all-gather of every rank's end state);
 * frames_per_rank: HOST int64 [world]; d_carry: DEVICE float [pairs][graphs][P][2] (out). */
sgz_status sgz_decay_fold_carry(sgz_plan *plan, const float *d_aggs, const int64_t *frames_per_rank,
                                uint32_t world, uint32_t rank, float *d_carry, void *stream);

/* The whole sharded render behind the ABI, on the host's own RCCL communicator (no torch, no Python): halo ncclSend / ncclRecv with
 * the neighbours (exactly the samples the last frames reach into the next rank's chunk), K_A, zero-carry K_B scan, ncclAllGather of
 * the end states, exact fold, K_B emit.  Rank r holds samples [r S, (r+1) S) of the stream in d_chunk (2*num_pairs channels, S =
 * chunk_samples, channel_stride >= S + halo_in: the halo is written behind the chunk); d_rgba receives this rank's columns
 * [local_frames][P][4].  Bit-identical to a single-device render of the concatenated stream.  (RSNT plans shard too: chunks of whole hops,
 * no halo; the resonators' linear recurrence is cut the same way -- every rank from rest, one all-gather of the resonators' end
 * states, the entering state folded in fp64 and added to every frame before the window kernel -- and the frames then equal a single
 * device's to within the bar its own chained frames are held to, not bit for bit: sharded.hip renderShardedResonator.)  nccl_comm: an ncclComm_t (RCCL is
 * bound with dlopen at first use; sgz_comm_* are conveniences for hosts that do not link RCCL themselves: sgz_comm_unique_id on
 * one rank, the 128 bytes handed to every rank by the host's own means, sgz_comm_create on all). */
sgz_status sgz_comm_unique_id(uint8_t out[128]);
sgz_status sgz_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, void **comm);
void       sgz_comm_destroy(void *comm);
sgz_status sgz_shard_layout(const sgz_plan *plan, uint32_t rank, uint32_t world, size_t chunk_samples, uint64_t *local_frames,
                            uint64_t *first_frame, uint64_t *halo_in, uint64_t *halo_out);
sgz_status sgz_spectrogram_render_sharded(sgz_plan *plan, void *nccl_comm, uint32_t rank, uint32_t world, float *d_chunk,
                                          size_t channel_stride, size_t chunk_samples, uint8_t *d_rgba, uint64_t *local_frames,
                                          void *stream);
/* The same render on the caller's own transport: the three collectives of the protocol as plain functions (counts in floats, device
 * pointers, 0 = success).  Each is ordered on `stream` the way ncclSend / ncclRecv / ncclAllGather are -- work queued on the stream
 * before the call is visible to it, work queued after it sees its result; an implementation may simply synchronise the stream and
 * move the data on the host.  Between group_begin and group_end (both optional) the send and the recv of one exchange are issued
 * back to back and must not deadlock on each other; abort (optional) is called when this rank fails while its peers may already be
 * waiting in a collective (ncclCommAbort on RCCL).  sgz_spectrogram_render_sharded is this call on RCCL. */
typedef struct sgz_transport {
    void *ctx;
    int (*send)(void *ctx, const float *d_buf, size_t count, uint32_t peer, void *stream);
    int (*recv)(void *ctx, float *d_buf, size_t count, uint32_t peer, void *stream);
    int (*allgather)(void *ctx, const float *d_send, float *d_recv /*[world][count]*/, size_t count, void *stream);
    int (*group_begin)(void *ctx);
    int (*group_end)(void *ctx);
    void (*abort)(void *ctx);
} sgz_transport;
sgz_status sgz_spectrogram_render_sharded_on(sgz_plan *plan, const sgz_transport *transport, uint32_t rank, uint32_t world, float *d_chunk,
                                             size_t channel_stride, size_t chunk_samples, uint8_t *d_rgba, uint64_t *local_frames,
                                             void *stream);
/* The same protocol for a host that drives several GPUs from ONE process (a thread per rank, no RCCL): an sgz_transport whose three
 * collectives are hipMemcpyPeerAsync copies over xGMI, each enqueued by the receiving rank on its own stream behind an event of the
 * sender's -- stream-ordered like their RCCL counterparts, and no host thread ever waits for a GPU (only for its peer to have enqueued).
 * devices[r] = HIP device of rank r (ranks may share a device: the copies are then plain device copies, which is how the tests run
 * it on one GPU).  One group per set of ranks; every rank takes its own transport from it and calls
 * sgz_spectrogram_render_sharded_on from its own thread with that device current.  *ctx_storage is released with
 * sgz_peer_transport_release when the rank is done. */
typedef struct sgz_peer_group sgz_peer_group;
sgz_status sgz_peer_group_create(uint32_t world, const int *devices /*[world]*/, sgz_peer_group **out);
void       sgz_peer_group_destroy(sgz_peer_group *group);
sgz_status sgz_peer_transport(sgz_peer_group *group, uint32_t rank, sgz_transport *out, void **ctx_storage);
void       sgz_peer_transport_release(void *ctx_storage);

/* ------------------------------------------------------------------------------------------------
 * Real-time per-block path: replaces Spectrum::ProcessorShell::onStreamAudio (SpectrumDSP.cpp:210-216)
 * -> AudioDispatcher::dispatch (:63-108) and the consumer side Spectrum::renderColourSpectrum's
 * frameQueue.popElement (SpectrumRendering.cpp:696-721) -- plus the two steps before the path: the additive
 * channel routing of MixGraphListener::deliver (Source/Common/MixGraphListener.cpp:247-334, sgz_spectrum_set_mix) and the audio
 * history ring, which lives in HBM (mirrored: the transform reads its window in place).  One producer thread (push) and one
 * consumer thread (pop_column, line_results, configure, set_mix, clear_state) may run concurrently.  push never waits for the GPU
 * and allocates nothing: when the GPU is several blocks behind, or a configure is in progress, it returns SGZ_BUSY and the block
 * is not taken.  At most 131072 samples per push.
 */
typedef struct sgz_spectrum sgz_spectrum;
sgz_status sgz_spectrum_create(const sgz_spectrum_config *cfg, sgz_spectrum **out);
void       sgz_spectrum_destroy(sgz_spectrum *s);
sgz_status sgz_spectrum_configure(sgz_spectrum *s, const sgz_spectrum_config *cfg);   /* handleFlagUpdates */
/* onStreamAudio(ctx, float** buffer, numChannels, numSamples) */
sgz_status sgz_spectrum_push(sgz_spectrum *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples);
/* frameQueue.popElement -> RGBA8 column of P pixels; SGZ_EMPTY when none is ready; SGZ_EINVAL on a LINE_GRAPH handle (display_mode 0 --
 * what a zero-initialised sgz_spectrum_config selects, as in the reference's enum -- produces no columns: sgz_spectrum_render_lines) */
sgz_status sgz_spectrum_pop_column(sgz_spectrum *s, uint8_t *rgba /*4*P*/, uint32_t *axis_points);
/* Display hand-off without the host: instead of popping columns and uploading each with
 * oglImage.updateSingleColumn (SpectrumRendering.cpp:696-721, :742-744), bind a device image of P rows x `columns` RGBA8 texels and
 * let sgz_spectrum_flush_columns (consumer thread, in place of the pop loop) scatter every ready column into it at
 * x = framePixelPosition, wrapping at `columns`; *first_column / *count name the texel columns written by this call (SGZ_EMPTY: none).
 *   sgz_spectrum_bind_image     caller-owned DEVICE memory (any mapped interop resource); d_image = NULL unbinds
 *   sgz_spectrum_create_image   the library allocates the image (whole 2 MiB blocks, see sgz_export_alloc) and exports it as a dma-buf
 *                               fd (the caller closes it): an MI355X has no graphics engine, the GL / Vulkan context of the display
 *                               GPU imports the fd (EXT_memory_object_fd, EGL_EXT_image_dma_buf_import; import size = pitch * P
 *                               rounded up to 2 MiB); dmabuf_fd may be NULL
 *   sgz_spectrum_bind_gl_buffer an OpenGL buffer object (e.g. a pixel-unpack buffer the texture is updated from) of a context that is
 *                               current on this thread and lives on the same device: hipGraphicsGLRegisterBuffer; flush_columns maps
 *                               and unmaps it around its writes.  tests/test_gpu_realtime.py test_gl_buffer_round_trip executes it where an
 *                               EGL surfaceless context can be made current; on the MI355X boxes this library is developed on it cannot (no
 *                               display engine; the image has libGL / GLX, which needs an X server, but neither libEGL nor libgbm) and the
 *                               test skips with the loader's error: there the call is only known to fail with a status
 * A configure drops the binding (the image height is the axis size). */
sgz_status sgz_spectrum_bind_image(sgz_spectrum *s, void *d_image, uint32_t columns, size_t pitch_bytes);
sgz_status sgz_spectrum_create_image(sgz_spectrum *s, uint32_t columns, void **d_image, size_t *pitch_bytes, int *dmabuf_fd);
sgz_status sgz_spectrum_bind_gl_buffer(sgz_spectrum *s, unsigned int gl_buffer, uint32_t columns, size_t pitch_bytes);
sgz_status sgz_spectrum_flush_columns(sgz_spectrum *s, uint32_t *first_column, uint32_t *count);
/* lineGraphs[graph].getResults(P) for pair `pair`: float2 [P] (TransformPair.h:72-76).  COLOUR_SPECTRUM: the results of the newest
 * frame whose copy has reached the host (a pinned triple buffer the producer's stream fills: this call waits for nothing and never
 * touches the producer's stream); LINE_GRAPH: the results of the last sgz_spectrum_render_lines. */
sgz_status sgz_spectrum_line_results(sgz_spectrum *s, uint32_t pair, uint32_t graph, float *out /*2*P*/);
/* DisplayMode::LineGraph, once per video frame on the render (consumer) thread: Spectrum::vectorGLRendering's
 *     pair.prepareTransform(constant, views) -> doTransform -> mapToLinearSpace -> postProcessStdTransform
 * for every pair (Source/Spectrum/SpectrumRendering.cpp:617-635; whole-ring prepareTransform TransformDSP.inl:39-231): the W newest
 * samples of the device ring are transformed and mapped, BOTH graphs' peak-decay filters advance once, and lineGraphs[k].results of every
 * pair come back: out = float2 [pairs][graphs][P] (what renderTransformAsGraph reads, SpectrumRendering.cpp:823-891).  RSNT: the windowed
 * state of the resonators as of the last pushed block (mapToLinearSpace's RSNT branch, :1103-1133).
 * poles: the graphs' decay for THIS video frame (the reference derives it from openGLDeltaTime(), Spectrum.cpp:388-394), or NULL =
 * the configured ones.  Waits for its own result (it is the render thread's call); never delays push.  SGZ_EINVAL on a
 * COLOUR_SPECTRUM handle. */
sgz_status sgz_spectrum_render_lines(sgz_spectrum *s, const float *poles /*[SGZ_NUM_GRAPHS] or NULL*/, float *out /*[pairs][graphs][P][2]*/);
/* Handle switches (consumer thread, between create / configure and the first push; a configure keeps them):
 *   SGZ_RT_OPT_STRICT_REFERENCE_QUIRKS  0 (default): ideal STFT framing -- a frame fires every `hop` samples wherever that falls inside a
 *       host block.  1: audioEntryPoint as written (TransformDSP.inl:1165-1211): every frame of one callback is prepared
 *       from the history BEFORE the callback plus the first min(availableSamples, W) samples of the UN-offset block (Q1: a 512-sample
 *       block at hop 200 yields the same window twice), and with an audio history longer than the window the frame is
 *       `history - W` samples short and zero-padded (Q2, :245-257).  Identical to the default whenever blocks divide the hop.
 *   SGZ_RT_OPT_AUDIO_HISTORY            the reference stream's audioHistorySize in samples (>= W; default and 0 = W, what
 *       Spectrum.cpp:472-477 asks for).  Read in strict mode only. */
#define SGZ_RT_OPT_STRICT_REFERENCE_QUIRKS 1u
#define SGZ_RT_OPT_AUDIO_HISTORY 2u
sgz_status sgz_spectrum_set_option(sgz_spectrum *s, uint32_t option, uint64_t value);
sgz_status sgz_spectrum_clear_state(sgz_spectrum *s);                                   /* clearAudioState, TransformPair.h:177-184 */
/* MixGraphListener::deliver's routing: destination channel d (of the 2*num_pairs the transform sees) = the sum of the source channels
 * c with matrix[d * num_sources + c] != 0, added in ascending c onto a cleared row (copyFromHead<true> into matrix.clear()'ed
 * rows).  push then takes num_sources channels.  Default: identity over 2*num_pairs sources. */
sgz_status sgz_spectrum_set_mix(sgz_spectrum *s, uint32_t num_sources, const uint8_t *matrix /*[2*num_pairs][num_sources]*/);
/* columns dropped because the queue was full (SpectrumDSP.cpp:185-186) and pushes refused with SGZ_BUSY, since create */
sgz_status sgz_spectrum_stats(sgz_spectrum *s, uint64_t *dropped_columns, uint64_t *refused_pushes);
/* (sgz_spectrum_backlog and sgz_spectrum_stats read atomics: a UI thread may poll them while the audio thread pushes.)
 * push never waits and never leaves a hole in the stream: a block the GPU is not ready for (all 8 staging slots in flight) waits in a
 * host FIFO -- one second of audio deep, like the reference's cpl::AudioStream in front of its listeners (PluginProcessor.cpp:195-198,
 * MixGraphListener.cpp:336-387) -- and is enqueued, in order, by the next push that finds a slot free.  SGZ_BUSY is returned only when
 * that FIFO is full (or a reconfiguration holds the handle).  deferred_blocks: blocks that ever waited there; waiting_now: its depth.
 * The same FIFO sits in front of sgz_scope_push and sgz_vector_push. */
sgz_status sgz_spectrum_backlog(sgz_spectrum *s, uint64_t *deferred_blocks, uint32_t *waiting_now);
/* enqueue whatever still waits in that FIFO (a stream that ended, a test): may wait for the GPU, so NOT for the audio thread -- and
 * not while the audio thread pushes (a push that finds the handle held is refused) */
sgz_status sgz_spectrum_flush(sgz_spectrum *s);
/* the hipStream_t the handle enqueues its work on (so that a host can order its own device work behind the handle's) */
void      *sgz_spectrum_stream(sgz_spectrum *s);
/* the frequency tracker on the newest window of pair `pair` (consumer thread): transforms the device ring's current window and runs
 * sgz_stage_track_peak's search on it */
sgz_status sgz_spectrum_track_peak(sgz_spectrum *s, uint32_t pair, double mouse_fraction, sgz_peak *out);
/* sgz_track_peak_lines on the handle's newest line results of (pair, graph) -- what sgz_spectrum_line_results would return now
 * (Complex mode, RSNT and the LineMain / LineSecond graphs: SpectrumRendering.cpp:300-377) */
sgz_status sgz_spectrum_track_peak_lines(sgz_spectrum *s, uint32_t pair, uint32_t graph, double mouse_fraction, sgz_line_peak *out);
/* parity hook: the W newest samples of destination channel `channel`, exactly the range of the device ring a frame firing now
 * would transform (call it from the producer's thread, or with the producer idle) */
sgz_status sgz_spectrum_history(sgz_spectrum *s, uint32_t channel, float *out /*W*/);

/* ------------------------------------------------------------------------------------------------
 * Oscilloscope: Lanczos-10 per-point resampler (drawWavePlot, OscilloscopeRendering.cpp:790-891),
 * zero-crossing trigger (ZeroCrossingProcessor::process, StreamPreprocessing.h:315-349) and the peak
 * envelope (runPeakFilter, OscilloscopeDSP.inl:713-886).
 */
typedef struct sgz_scope_view {
    double   window_size;       /* state.effectiveWindowSize, samples      */
    double   left, right;       /* state.viewOffsets[Left], [Right]        */
    double   rendering_scale;   /* oglc->getRenderingScale()               */
    uint32_t width;             /* getWidth(), pixels                      */
    uint32_t _pad;
} sgz_scope_view;
size_t     sgz_scope_num_points(const sgz_scope_view *view);
/* d_ring: DEVICE fp32 front buffer in time order (index 0 at the stream cursor), len samples, `channels`
 * buffers at d_ring + c*stride.  d_xy: DEVICE float2 [channels][points] = the (x, y) of addVertex(x, y, 0). */
sgz_status sgz_scope_lanczos_device(const sgz_scope_view *view, const float *d_ring, size_t len,
                                    size_t stride, uint32_t channels, float *d_xy, void *stream);
typedef struct sgz_zero_crossing_state {
    double   state;
    double   threshold;
    uint64_t steady_clock;
    uint64_t cross_origin;
    uint64_t count;
    int32_t  armed;
    int32_t  _pad;
} sgz_zero_crossing_state;
/* d_a/d_b: DEVICE trigger-pair channels for this block; d_triggers: DEVICE uint64 [max_triggers] absolute
 * sample indices (the values peaks.push receives); *num_triggers and *st are updated on the host
 * (this call synchronises the stream: the trigger list is consumed by host logic). */
sgz_status sgz_scope_zero_crossing_device(sgz_zero_crossing_state *st, uint32_t osc_mode, const float *d_a,
                                          const float *d_b, size_t n, uint64_t *d_triggers,
                                          size_t max_triggers, size_t *num_triggers, void *stream);
/* peak envelope over `channels` windows of n samples; env (host, in/out, [channels]); returns gain */
sgz_status sgz_peak_filter_device(const float *d_ch, size_t stride, uint32_t channels, size_t n,
                                  uint32_t lanes, double coeff_pow, double *env, double *gain, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Oscilloscope real-time handle: replaces Oscilloscope::ProcessorShell::onStreamAudio (Source/Oscilloscope/Oscilloscope.h:293) ->
 * StreamState::audioEntryPoint (OscilloscopeDSP.inl:401-424) on the audio thread, and on the render thread
 * Oscilloscope::runPeakFilter (OscilloscopeDSP.inl:713-886) and drawWavePlot (OscilloscopeRendering.cpp:551-891, the Linear and
 * Lanczos branches) -> the (x, y, z) + colour stream PrimitiveDrawer::addVertex / addColour receive.
 * The trigger detector, TriggeringProcessor::processMutating's window selection (StreamPreprocessing.h:79-206), the back / front
 * rings (ChannelData.h) and the envelope all live in HBM; push stages the block -- an idle GPU starts on it at once, a busy one takes
 * everything that arrived meanwhile in ONE staged copy + ONE launch (the blocks keep their boundaries) --, and push never waits for
 * the GPU (SGZ_BUSY instead).  One producer thread (push), one consumer thread (everything else).
 * trigger mode Spectral (sgz_scope_analyse = calculateFundamentalPeriod + calculateTriggeringOffset,
 * OscilloscopeDSP.inl:62-308, on the device ring) and the per-sample frequency colouring of audioProcessing (:445-647: 3-band
 * Linkwitz-Riley split -> smoothed band energies -> RGB, kept in colour rings beside the audio rings and swapped with them).
 * The remaining modes too: EnvelopeHold (PeakHoldProcessor, StreamPreprocessing.h:270-310, inside the ingest kernel), Window
 * (sgz_scope_set_transport), customTrigger, and the None / Rectangular interpolations of drawWavePlot. */
typedef struct sgz_scope_config {
    double   sample_rate;
    double   window_size;        /* state.effectiveWindowSize in samples (fractions allowed)                     */
    uint32_t num_channels;       /* even, 2..64                                                                  */
    uint32_t trigger_mode;       /* SGZ_TRIG_* (Window: see sgz_scope_set_transport)                             */
    uint32_t channel_mode;       /* SGZ_OSC_* (OscChannels): trigger mix, envelope mix                           */
    uint32_t envelope_mode;      /* SGZ_ENV_*: RMS runs in push (audioProcessing), PEAK_DECAY in sgz_scope_peak_filter */
    uint32_t interpolation;      /* SGZ_SUBSAMPLE_*: None = the Linear vertex list, drawn as GL_POINTS (dotSamples,
                                    OscilloscopeRendering.cpp:652-700); Rectangular = two vertices per sample (:746-789) */
    uint32_t max_block;          /* longest block push will be given (0: 8192)                                   */
    double   trigger_threshold;  /* content->triggerThreshold                                                    */
    double   trigger_channel;    /* content->triggeringChannel, 1-based (calculateTriggerIndices)                */
    double   envelope_window;    /* content->envelopeWindow normalised value = seconds             */
    uint8_t  colours[64][4];     /* per channel: filterStates.channels[c].defaultKey as RGBA8                    */
    /* Spectral triggering (all ignored in the other modes) */
    double   trigger_hysteresis;   /* content->triggerHysteresis, 0..1                                           */
    double   trigger_phase_offset; /* content->triggerPhaseOffset in degrees                                     */
    /* frequency colouring */
    uint32_t colour_by_frequency;  /* state.colourChannelsByFrequency: colours computed in push, per-vertex colours from the rings */
    float    frequency_colouring_blend;   /* content->frequencyColouringBlend, 0..1                              */
    double   colour_smoothing_ms;  /* content->colourSmoothing (transformed value, milliseconds)                 */
    float    band_colours[3][3];   /* content->lowColour / midColour / highColour as float r, g, b               */
    /* Spectral triggering on a frequency the user names (state.customTrigger / customTriggerFrequency, OscilloscopeDSP.inl:71-81) */
    uint32_t custom_trigger;
    double   custom_trigger_frequency;    /* Hz, in (0, sample_rate / 2)                                          */
} sgz_scope_config;
/* Oscilloscope::triggerState after analyseAndSetupState's first two steps (Oscilloscope.h:176-196) */
typedef struct sgz_trigger_state {
    uint64_t record_index;         /* triggerState.record: the winning bin, its magnitude and its fractional offset */
    double   record_value, record_offset;
    double   fundamental;          /* Hz, >= 5 */
    double   cycle_samples;        /* sampleRate / fundamental (0 outside Spectral mode)                          */
    double   sample_offset;        /* samples                                                                      */
    double   phase;                /* radians, [0, tau)                                                            */
    uint64_t ring_size;            /* ChannelData::resizeAudioStorage's size for this frame: the ring drawWavePlot wraps in */
} sgz_trigger_state;
typedef struct sgz_scope sgz_scope;
sgz_status sgz_scope_create(const sgz_scope_config *cfg, sgz_scope **out);
void       sgz_scope_destroy(sgz_scope *s);
/* handleFlagUpdates -> TriggeringProcessor::setSettings (Oscilloscope.cpp:310).  A changed ceil(window) resizes (and clears) the rings. */
sgz_status sgz_scope_configure(sgz_scope *s, const sgz_scope_config *cfg);
/* onStreamAudio(ctx, float** buffer, numChannels, numSamples); the steady clock is the running count of pushed samples */
sgz_status sgz_scope_push(sgz_scope *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples);
sgz_status sgz_scope_flush(sgz_scope *s);      /* as sgz_spectrum_flush */
/* Handle switch (consumer thread, between create / configure and the first push; the library reads no environment variable):
 *   SGZ_RT_OPT_DEFER_SUBMIT  0 (default): a pushed block goes to the GPU at once when nothing of the handle is in flight, otherwise it
 *       joins the open batch, which the next submission takes in ONE launch.  1: every block waits for a full batch or a reader
 *       (flush on read) -- every launch a multi-callback one.  Results are identical; the tests pin the batched paths down with it.
 *   SGZ_RT_OPT_PARK_PUSHES   0 (default).  1: every pushed block is parked in the handle's host FIFO -- where a block goes whose push finds the
 *       render thread submitting at that moment, or no staging slot free -- until the next read (flush on read covers the FIFO), push or
 *       flush hands it on.  Results are identical; the tests reproduce a stopped transport with it. */
#define SGZ_RT_OPT_DEFER_SUBMIT 3u
#define SGZ_RT_OPT_PARK_PUSHES 4u
sgz_status sgz_scope_set_option(sgz_scope *s, uint32_t option, uint64_t value);
/* TriggeringMode::Window draws the window at the host transport's phase: position_in_samples = cs.transportPosition =
 * playhead.getPositionInSamples() + numSamples of the newest block (OscilloscopeDSP.inl:706; OscilloscopeRendering.cpp:588-592,
 * :798-801).  Any thread, any time (one atomic store); ignored by the other modes. */
sgz_status sgz_scope_set_transport(sgz_scope *s, int64_t position_in_samples);
/* runPeakFilter once per rendered frame: delta_time = openGLDeltaTime(), lanes = the SIMD width whose tail the reference drops
 * (8 = AVX); *auto_gain = state.autoGain (optional; reading it waits for the kernel) */
sgz_status sgz_scope_peak_filter(sgz_scope *s, double delta_time, uint32_t lanes, double *auto_gain);
/* envelopeGain of the RMS mode and the per-channel envelope states (either may be NULL) */
sgz_status sgz_scope_gains(sgz_scope *s, double *envelope_gain, float *envelopes /*num_channels*/);
/* Once per rendered frame, before sgz_scope_vertices: calculateFundamentalPeriod + calculateTriggeringOffset
 * (OscilloscopeDSP.inl:62-308) for the trigger evaluator (evaluator / channel as in sgz_scope_vertices).  Spectral mode: one kernel
 * (8192-point fp64 transform of the newest samples in LDS, the harmonic peak pick with hysteresis, the median of 8, the Goertzel
 * phase) on the device ring; the state is read back (this call waits) and kept for the vertex calls.  Other modes: cycle_samples = 0
 * and the mode's fixed sample_offset, no GPU work.  The Spectral ring keeps the most the reference can ask for
 * ((size_t)(0.5 + sampleRate / 5 + ceil(window)) samples, the 5 Hz floor) and every read wraps in `ring_size`, the size
 * resizeAudioStorage gives the reference's ring for this frame (ChannelData.h:107-128), counted back from the newest sample. */
sgz_status sgz_scope_analyse(sgz_scope *s, uint32_t evaluator, uint32_t channel, sgz_trigger_state *out);
size_t     sgz_scope_vertex_count(const sgz_scope *s, const sgz_scope_view *view);
/* One evaluator's line strip.  evaluator: SGZ_OSC_LEFT / RIGHT (channel `channel` / `channel` + 1) or SGZ_OSC_MID / SIDE (0.5 (l +- r)
 * of the pair at `channel`); view->window_size is ignored (the stream's is used).  xyz: float3 per vertex, rgba: RGBA8 per vertex
 * (may be NULL); *count: in = capacity of the buffers in vertices, out = vertices written.  Lanczos below one pixel per sample
 * falls back to Linear like the reference (OscilloscopeRendering.cpp:575-578): x is then the sample index (sample space).
 * Host buffers that are pinned (hipHostMalloc / hipHostRegister) are written by the DMA engine directly; pageable ones go through a
 * pinned bounce buffer and a host copy (the same holds for sgz_vector_vertices / _all). */
sgz_status sgz_scope_vertices(sgz_scope *s, const sgz_scope_view *view, uint32_t evaluator, uint32_t channel, float *xyz,
                              uint8_t *rgba, uint32_t *count);
/* Several evaluators' line strips of one rendered frame (the reference draws them one after the other inside one paint,
 * OscilloscopeRendering.cpp drawWavePlot per channel): item k is sgz_scope_vertices(s, view, evaluators[k], channels[k], xyz[k],
 * rgba ? rgba[k] : NULL, &counts[k]), but the kernels are enqueued back to back and the call waits ONCE.  Same results, same buffer
 * rules; any failing item fails the call (counts[k] holds the required size where a buffer was too small). */
sgz_status sgz_scope_vertices_all(sgz_scope *s, const sgz_scope_view *view, uint32_t items, const uint32_t *evaluators, const uint32_t *channels,
                                  float *const *xyz, uint8_t *const *rgba, uint32_t *counts);
/* (the buffers of sgz_scope_vertices_all may also be DEVICE memory -- all of them --: the kernels then write HBM, one wait, nothing crosses PCIe) */
/* the hipStream_t the handle enqueues its work on (so that a host can order or time its own device work against the handle's) */
void      *sgz_scope_stream(sgz_scope *s);
/* The same stream into DEVICE buffers -- a mapped vertex buffer object, or memory from sgz_export_alloc that the display GPU's GL /
 * Vulkan imported -- without the D2H copy.  In place when the call returns. */
sgz_status sgz_scope_vertices_device(sgz_scope *s, const sgz_scope_view *view, uint32_t evaluator, uint32_t channel, float *d_xyz,
                                     uint8_t *d_rgba, uint32_t *count);
/* parity hooks: front buffer memory of one channel (begin()) + its write cursor; TriggeringProcessor counters
 * {frontOrigin, bufferedSamples, oldPeak, currentPeak, steadyClock, peaks.size(), isWorkingOnPeak, swaps} */
sgz_status sgz_scope_front(sgz_scope *s, uint32_t channel, float *out /*size*/, uint32_t *size, uint32_t *cursor);
/* the colour ring beside it: aux = 0 colourData, 1 auxColourData (Mid at even, Side at odd channels); RGBA8 words */
sgz_status sgz_scope_front_colours(sgz_scope *s, uint32_t channel, uint32_t aux, uint8_t *out /*4*size*/);
sgz_status sgz_scope_debug_state(sgz_scope *s, uint64_t out[8]);

/* ------------------------------------------------------------------------------------------------
 * Vectorscope: polar transform (drawPolarPlot, VectorscopeRendering.cpp:500-746) and the audio-thread
 * one-pole filters (Processor::audioProcessing, Vectorscope.cpp:268-377).
 */
/* d_xyz: DEVICE float3 [pairs][n]; pair p uses channels 2p, 2p+1 of d_planar */
sgz_status sgz_vector_polar_device(const float *d_planar, size_t stride, uint32_t pairs, size_t n,
                                   uint32_t lanes, float *d_xyz, void *stream);
typedef struct sgz_vector_filters { float env[2]; float balance[2][2]; float phase[2]; } sgz_vector_filters;
sgz_status sgz_vector_audio_processing_device(sgz_vector_filters *f, const float *d_left, const float *d_right,
                                              size_t n, uint32_t lanes, float envelope_coeff, float stereo_coeff,
                                              float second_speed, int env_mode, float *gain_out, void *stream);

/* Vectorscope real-time handle: replaces VectorScope::Processor::onStreamAudio -> audioProcessing (Source/Vectorscope/Vectorscope.h:141,
 * Vectorscope.cpp:268-392) and the cpl::AudioStream history the renderer reads, and on the render thread VectorScope::runPeakFilter
 * (VectorscopeRendering.cpp:826-889) and drawPolarPlot (:500-746) for every channel pair -> the (x, y, z) + (r, g, b) stream
 * PrimitiveDrawer::addVertex / addColour receive.  History ring, filter states and gain live in HBM; push = (batched, as sgz_scope_push) one staged copy + one
 * launch and never waits for the GPU (SGZ_BUSY instead).  One producer thread (push), one consumer thread (everything else). */
typedef struct sgz_vector_config {
    double   sample_rate;
    uint32_t num_channels;       /* even, 2..64; pair p = channels 2p, 2p+1; the filters listen to channels 0, 1       */
    uint32_t window_size;        /* audio history in samples = vertices per pair                                       */
    uint32_t envelope_mode;      /* SGZ_ENV_*: RMS updates the gain in push, PEAK_DECAY in sgz_vector_peak_filter      */
    uint32_t lanes;              /* SIMD width of the reference build (8 = AVX): tails it drops / handles in scalar code */
    uint32_t fade_history;       /* state.fadeHistory: colours fade with age (VectorscopeRendering.cpp:637-746)         */
    uint32_t max_block;          /* longest block push will be given (0: 8192)                                          */
    double   envelope_window;    /* seconds (content->envelopeWindow normalised)                                        */
    double   stereo_window;      /* seconds (content->stereoWindow normalised)                                          */
    float    colours[32][3];     /* per pair: the waveform colour as getFloatRed / Green / Blue                         */
} sgz_vector_config;
typedef struct sgz_vector sgz_vector;
sgz_status sgz_vector_create(const sgz_vector_config *cfg, sgz_vector **out);
void       sgz_vector_destroy(sgz_vector *s);
sgz_status sgz_vector_configure(sgz_vector *s, const sgz_vector_config *cfg);
sgz_status sgz_vector_push(sgz_vector *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples);
sgz_status sgz_vector_flush(sgz_vector *s);    /* as sgz_spectrum_flush */
sgz_status sgz_vector_set_option(sgz_vector *s, uint32_t option, uint64_t value);   /* SGZ_RT_OPT_DEFER_SUBMIT / SGZ_RT_OPT_PARK_PUSHES, as sgz_scope_set_option */
sgz_status sgz_vector_peak_filter(sgz_vector *s, double delta_time, double *envelope_gain /*optional: reading it waits*/);
sgz_status sgz_vector_filters_get(sgz_vector *s, sgz_vector_filters *filters, double *envelope_gain);
/* xyz: float3 [window_size], rgb: float3 [window_size] or NULL; *count: in = capacity in vertices, out = window_size.  Vertex
 * order = the reference's: the older section of the ring ([cursor, size)) first, then [0, cursor). */
sgz_status sgz_vector_vertices(sgz_vector *s, uint32_t pair, float *xyz, float *rgb, uint32_t *count);
sgz_status sgz_vector_vertices_device(sgz_vector *s, uint32_t pair, float *d_xyz, float *d_rgb, uint32_t *count);   /* DEVICE buffers */
/* every pair's stream with ONE wait for the GPU (the render thread draws all pairs of a frame, VectorscopeRendering.cpp:253-276):
 * xyz / rgb: float3 [num_channels / 2][window_size]; *count: in = capacity PER PAIR, out = window_size */
sgz_status sgz_vector_vertices_all(sgz_vector *s, float *xyz, float *rgb, uint32_t *count);
/* (xyz / rgb of sgz_vector_vertices_all may also be DEVICE memory: the vertices stay in HBM) */
void      *sgz_vector_stream(sgz_vector *s);       /* as sgz_scope_stream */
/* parity hook: history ring memory of one channel + the write cursor */
sgz_status sgz_vector_history(sgz_vector *s, uint32_t channel, float *out /*window_size*/, uint32_t *size, uint32_t *cursor);

#ifdef __cplusplus
}
#endif
#endif /* SGZ_H */
