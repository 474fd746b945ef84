// realtime.hip -- the per-block real-time Spectrum path behind sgz_spectrum_push / sgz_spectrum_pop_column.
//
// Replaces Spectrum::ProcessorShell::onStreamAudio -> AudioDispatcher::dispatch (Source/Spectrum/SpectrumDSP.cpp:63-108,
// :210-216), TransformPair::audioEntryPoint's frame cadence (TransformDSP.inl:1165-1211: processedSamplesSinceLastFrame /
// sampleBufferSize), the consumer side of the frameQueue (SpectrumRendering.cpp:696-721) -- and the two steps BEFORE the path
// (SURVEY.md 8(f) #2): MixGraphListener::deliver's additive routing of source channels into destination ports
// (Source/Common/MixGraphListener.cpp:247-334) and the cpl::AudioStream history ring prepareTransform gathers its two segments
// from (TransformDSP.inl:65-88, :234-484).
//
// Device-resident ring.  Every destination channel owns a MIRRORED ring in HBM: capacity `cap`, every sample stored twice, at
// p and p + cap.  Any window of <= cap samples is therefore one contiguous range of memory, whatever the write position: K_A's
// load stage reads a frame's W newest samples IN PLACE with its ordinary linear addressing -- no two-segment gather, no
// compaction copies, no modular arithmetic in the hot kernel; the price is that the ingest kernel writes each (tiny) block twice.
// The ingest kernel applies the mix matrix on the way in: destination d = sum over the source channels c routed to it, in
// ascending c, starting from the cleared matrix (0 + a + b ...: the reference's copyFromHead<true> into matrix.clear()'ed rows).
//
// Framing (DisplayMode::ColourSpectrum) is by default the ideal STFT framing: a frame fires every `hop` samples and covers the W samples
// that end at the firing point; all frames that fire inside one staged piece go through K_A and K_B as ONE launch each (their windows
// are hop-spaced ranges of the same ring).  sgz_spectrum_set_option(SGZ_RT_OPT_STRICT_REFERENCE_QUIRKS, 1) reproduces
// audioEntryPoint as written instead: the frames of one callback all read the history BEFORE the callback plus
// the first min(availableSamples, W) samples of the un-offset block -- in the mirrored ring that is still one contiguous window, it just
// ends `stop` samples into the block instead of where the hop fell -- and a history longer than the window shortens the frame (a
// small gather kernel builds those).
//
// DisplayMode::LineGraph (the reference's default): push only feeds the ring (RSNT: and advances the resonators over the block,
// TransformDSP.inl:1206-1209); sgz_spectrum_render_lines, on the consumer thread once per video frame, transforms the ring's newest
// window for every pair and advances both graphs' filters once (SpectrumRendering.cpp:617-635).  It runs on the consumer's own Plan
// (`trackPlan`: per-launch scratch is per Plan) and reads the ring without a lock: the producer publishes how far it is ABOUT to
// write before it enqueues an ingest kernel, the consumer checks that figure after it has enqueued its transform, and repeats the
// (idempotent) transform in the one-in-a-million case where its window could have been overtaken.
//
// Threading: one producer thread (push) and one consumer thread (pop_column / line_results / configure / clear_state).
// push never waits for the GPU: staging slots and column slots are checked with hipEventQuery / atomics, allocations and LDS
// grants happen in create / configure (a warm-up render of the largest batch), and a push that finds the GPU too far behind
// returns SGZ_BUSY without having consumed anything.
//
// Display hand-off without the host (replaces oglImage.updateSingleColumn per popped frame,
// SpectrumRendering.cpp:696-721, :742-744).  The column queue also exists in HBM; sgz_spectrum_flush_columns (consumer thread)
// scatters every ready column into a bound device image [P rows][pitch] at x = framePixelPosition -- exactly the texels
// updateSingleColumn would upload.  The image is (a) caller-owned device memory (sgz_spectrum_bind_image: any mapped interop
// resource), (b) allocated here and exported as a dma-buf fd (sgz_spectrum_create_image: the MI355X has no graphics engine, so the
// GL / Vulkan context lives on the display GPU and imports the fd with EXT_memory_object_fd / EGL_EXT_image_dma_buf_import), or
// (c) an OpenGL buffer object of a context on the same device, registered and mapped through HIP's GL interop
// (sgz_spectrum_bind_gl_buffer).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <hip/hip_gl_interop.h>

#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "rt_common.hpp"
#include "trace.hpp"

using namespace sgz;

namespace {
constexpr int kQueueDepth = 10;              // frameQueue(10), SpectrumDSP.cpp:47
constexpr uint32_t kPiece = 16384;           // samples per staged piece (a push is cut into pieces of at most this)
constexpr uint32_t kMaxSources = 64;

// dest[d][i] = sum_{c : mix[d][c]} src[c][i]  (ascending c, from 0), written at ring position (head + i) mod cap and + cap
__global__ void __launch_bounds__(256)
ringIngestKernel(const float *src, uint32_t n, uint32_t numSrc, const uint8_t *mix, float *ring, uint32_t cap, uint32_t numDst,
                 uint32_t head)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, d = blockIdx.y;
    if (i >= n || d >= numDst) return;
    float v = 0.f;
    for (uint32_t c = 0; c < numSrc; ++c)
        if (mix[d * numSrc + c]) v = v + src[size_t(c) * n + i];
    uint32_t p = head + i; if (p >= cap) p -= cap;
    float *r = ring + size_t(d) * 2 * cap;
    r[p] = v;
    r[p + cap] = v;
}

// Quirk Q2 (TransformDSP.inl:245-257), audio history `extra` samples longer than the window: the frame is
//     history[T - W + stop, T - extra) ++ block[0, stop) ++ `extra` zeros        (T = the block's first sample)
// -- the newest `extra` samples of the history are skipped and the window function is applied at the shifted positions.  src = ring
// position of sample T - W + stop (mirrored ring: contiguous); out [numDst][W].
__global__ void __launch_bounds__(256)
strictGatherKernel(const float *ring, uint32_t cap, uint32_t src, uint32_t W, uint32_t stop, uint32_t extra, float *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, d = blockIdx.y;
    if (i >= W) return;
    const float *r = ring + size_t(d) * 2 * cap + src;
    const uint32_t fromHistory = W - stop - extra;
    out[size_t(d) * W + i] = i < fromHistory ? r[i] : i < W - extra ? r[i + extra] : 0.f;
}

// texel (x, y) = column[y]: what updateSingleColumn(x, column) uploads into a P-row texture
__global__ void __launch_bounds__(256) columnScatterKernel(const uint32_t *column, uint8_t *image, size_t pitch, uint32_t x, uint32_t P)
{
    const uint32_t y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y < P) *reinterpret_cast<uint32_t *>(image + size_t(y) * pitch + size_t(x) * 4) = column[y];
}
}  // namespace

struct sgz_spectrum {
    Plan *plan = nullptr;
    Plan *trackPlan = nullptr;                 // the frequency tracker's own constant block + launch scratch: sgz_spectrum_track_peak runs on the consumer
                                               // thread while push launches K_A from the producer's -- they must not share a Plan's per-launch buffers
    std::mutex cfgMu;                 // configure (consumer thread) against push (producer, try_lock only)
    hipStream_t stream = nullptr;
    StageRing stage;
    Backlog backlog;                           // blocks waiting for a staging slot (rt_common.hpp)
    // mirrored rings [2C][2 cap]
    float *d_ring = nullptr;
    uint32_t cap = 0;
    // samples the producer has enqueued into the ring (`written`; the write position is written % cap) and is about to enqueue
    // (`planned`, stored BEFORE the ingest launch): the consumer-side transforms (tracker, line graph) read a window that ends at
    // `written` and verify against `planned` afterwards that no ingest enqueued in front of them can have reached it
    std::atomic<uint64_t> written{0}, planned{0};
    uint32_t sinceLast = 0;           // processedSamplesSinceLastFrame
    bool strictQuirks = false;        // SGZ_RT_OPT_STRICT_REFERENCE_QUIRKS
    uint64_t audioHistory = 0;        // SGZ_RT_OPT_AUDIO_HISTORY (0 = W)
    float *d_strict = nullptr;        // strict mode, history > W: one frame's gathered input [2C][W]
    // line graph (consumer thread): K_A's output for the ring's newest window, the results on the host
    float *d_lineMapped = nullptr;    // [C][sides][P]
    float *h_lineOut = nullptr;       // pinned [C][graphs][P][2]
    // colour spectrum: lineGraphs[k].results of the newest frame, copied to the host behind every batch (triple buffer, seqlock)
    static constexpr int kLineSlots = 3;
    float *h_lines = nullptr;         // pinned [kLineSlots][C][graphs][P][2]
    hipEvent_t lineEvents[kLineSlots] = {};
    LineSeqlock<kLineSlots> lineSeq;  // copies the producer has announced / enqueued (rt_lockfree.hpp; slot = (n - 1) % kLineSlots)
    std::atomic<uint64_t> deferredStat{0}; std::atomic<uint32_t> waitingStat{0};   // the backlog's counters for sgz_spectrum_backlog
    uint32_t maxFrames = 1;
    uint8_t *d_mix = nullptr;
    uint32_t numSources = 0;
    float *d_mapped = nullptr, *d_state = nullptr, *d_lines = nullptr, *d_linesBatch = nullptr;
    float *d_trackBins = nullptr; sgz_peak *d_peak = nullptr;     // frequency tracker: csf of the newest window [C][N + 1], its result
    uint8_t *d_colsBatch = nullptr;   // [maxFrames][P][4]
    uint8_t *h_cols = nullptr;        // pinned [kQueueDepth][P][4]
    hipEvent_t colEvents[kQueueDepth] = {};
    // SPSC column queue: the producer fills slot tail % depth and bumps tail, the consumer reads slot head % depth and bumps head
    ColumnQueue<kQueueDepth> colQ;    // rt_lockfree.hpp
    std::atomic<uint64_t> dropped{0}, busy{0};
    // display hand-off (consumer thread): device copy of the queue slots, the bound image, framePixelPosition
    uint8_t *d_colsQ = nullptr;       // [kQueueDepth][P][4]
    hipStream_t outStream = nullptr;
    uint8_t *d_image = nullptr; size_t imgPitch = 0; uint32_t imgColumns = 0, imgX = 0;
    bool imgOwned = false;
    hipGraphicsResource *glResource = nullptr;
};

static void unbindImage(sgz_spectrum *s)
{
    if (s->outStream) (void)hipStreamSynchronize(s->outStream);
    if (s->glResource) {
        (void)hipGraphicsUnregisterResource(s->glResource);          // (never left mapped: flush_columns maps and unmaps around its writes)
        s->glResource = nullptr;
    }
    if (s->imgOwned && s->d_image) (void)hipFree(s->d_image);
    s->d_image = nullptr; s->imgOwned = false; s->imgColumns = 0; s->imgPitch = 0; s->imgX = 0;
}

static void freeHandle(sgz_spectrum *s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    s->stage.release();
    s->backlog.release();
    for (float *p : {s->d_ring, s->d_mapped, s->d_state, s->d_lines, s->d_linesBatch, s->d_trackBins, s->d_strict, s->d_lineMapped}) if (p) (void)hipFree(p);
    if (s->h_lineOut) (void)hipHostFree(s->h_lineOut);
    if (s->h_lines) (void)hipHostFree(s->h_lines);
    for (auto &e : s->lineEvents) if (e) (void)hipEventDestroy(e);
    if (s->d_peak) (void)hipFree(s->d_peak);
    unbindImage(s);
    if (s->outStream) (void)hipStreamDestroy(s->outStream);
    if (s->d_colsQ) (void)hipFree(s->d_colsQ);
    if (s->d_colsBatch) (void)hipFree(s->d_colsBatch);
    if (s->d_mix) (void)hipFree(s->d_mix);
    if (s->h_cols) (void)hipHostFree(s->h_cols);
    for (auto &e : s->colEvents) if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s->plan;
    delete s->trackPlan;
    delete s;
}

static sgz_status uploadMix(sgz_spectrum *s, uint32_t numSources, const uint8_t *matrix)
{
    const uint32_t numDst = 2 * s->plan->C;
    std::vector<uint8_t> m(size_t(numDst) * numSources, 0);
    if (matrix) std::memcpy(m.data(), matrix, m.size());
    else for (uint32_t d = 0; d < numDst && d < numSources; ++d) m[size_t(d) * numSources + d] = 1;     // identity routing
    if (s->d_mix) { (void)hipFree(s->d_mix); s->d_mix = nullptr; }
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_mix), m.size()));
    SGZ_HIP(hipMemcpy(s->d_mix, m.data(), m.size(), hipMemcpyHostToDevice));
    s->numSources = numSources;
    // one second of audio may wait for the GPU (at least 32 staging pieces)
    if (!s->backlog.init(backlogFloats(numSources, s->plan->cfg.sample_rate, 8 * kPiece))) return fail(SGZ_ENOMEM, "out of memory (push backlog)");
    return s->stage.init(numSources, kPiece);
}

// builds everything for a configuration into the handle (the caller holds cfgMu, or the handle is not shared yet)
static sgz_status setup(sgz_spectrum *s, const sgz_spectrum_config *cfg)
{
    Plan *pl = new (std::nothrow) Plan();
    if (!pl) return fail(SGZ_ENOMEM, "out of memory");
    std::string err;
    sgz_status st;
    try { st = buildPlan(*cfg, *pl, err); }
    catch (const std::bad_alloc &) { st = SGZ_ENOMEM; err = "out of memory building the plan tables"; }
    if (st == SGZ_OK) st = uploadPlan(*pl, err);
    if (st != SGZ_OK) { delete pl; return fail(st, err); }
    Plan *tp = new (std::nothrow) Plan();
    if (!tp) { delete pl; return fail(SGZ_ENOMEM, "out of memory"); }
    try { st = buildPlan(*cfg, *tp, err); }
    catch (const std::bad_alloc &) { st = SGZ_ENOMEM; err = "out of memory building the plan tables"; }
    if (st == SGZ_OK) st = uploadPlan(*tp, err);
    if (st != SGZ_OK) { delete pl; delete tp; return fail(st, err); }
    if (!s->stream) SGZ_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    (void)hipStreamSynchronize(s->stream);
    delete s->plan;
    s->plan = pl;
    delete s->trackPlan;
    s->trackPlan = tp;
    Plan &p = *pl;
    const size_t nch = size_t(2) * p.C;
    for (float **q : {&s->d_ring, &s->d_mapped, &s->d_state, &s->d_lines, &s->d_linesBatch, &s->d_trackBins, &s->d_strict, &s->d_lineMapped}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    if (s->h_lineOut) { (void)hipHostFree(s->h_lineOut); s->h_lineOut = nullptr; }
    if (s->h_lines) { (void)hipHostFree(s->h_lines); s->h_lines = nullptr; }
    if (s->d_colsBatch) { (void)hipFree(s->d_colsBatch); s->d_colsBatch = nullptr; }
    if (s->h_cols) { (void)hipHostFree(s->h_cols); s->h_cols = nullptr; }
    if (s->d_colsQ) { (void)hipFree(s->d_colsQ); s->d_colsQ = nullptr; }
    unbindImage(s);                                        // the image's height is the axis size: a new configuration needs a new binding
    // a piece's frames read windows that end inside the piece: the ring must hold W + one piece (RSNT: a frame consumes the `hop`
    // samples that end with it) -- and a second piece of slack for the consumer thread's transforms of the newest window (see above)
    s->cap = ((isResonator(p) ? p.cfg.hop : p.W) + 2 * kPiece + 63u) & ~63u;
    s->maxFrames = kPiece / p.cfg.hop + 1;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_ring), nch * 2 * s->cap * sizeof(float)));
    SGZ_HIP(hipMemsetAsync(s->d_ring, 0, nch * 2 * s->cap * sizeof(float), s->stream));    // history starts as silence
    s->written.store(0); s->planned.store(0);
    s->sinceLast = 0;
    s->lineSeq.reset();
    s->deferredStat.store(0); s->waitingStat.store(0);
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_mapped), size_t(s->maxFrames) * p.C * p.sides * p.P * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_state), stateN * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_lines), stateN * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_linesBatch), size_t(s->maxFrames) * stateN * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_colsBatch), size_t(s->maxFrames) * p.P * 4));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_lineMapped), size_t(p.C) * p.sides * p.P * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_strict), nch * p.W * sizeof(float)));
    SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_lineOut), stateN * sizeof(float), hipHostMallocDefault));
    SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_lines), size_t(sgz_spectrum::kLineSlots) * stateN * sizeof(float), hipHostMallocDefault));
    std::memset(s->h_lineOut, 0, stateN * sizeof(float));
    std::memset(s->h_lines, 0, size_t(sgz_spectrum::kLineSlots) * stateN * sizeof(float));
    for (auto &e : s->lineEvents) if (!e) SGZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (p.cfg.channel_mode != SGZ_CH_PHASE && !isResonator(p)) SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_trackBins), size_t(p.C) * (size_t(p.N) + 1) * sizeof(float)));
    if (!s->d_peak) SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_peak), sizeof(sgz_peak)));
    SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_cols), size_t(kQueueDepth) * p.P * 4, hipHostMallocDefault));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_colsQ), size_t(kQueueDepth) * p.P * 4));
    if (!s->outStream) SGZ_HIP(hipStreamCreateWithFlags(&s->outStream, hipStreamNonBlocking));
    for (auto &e : s->colEvents) if (!e) SGZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->colQ.reset();
    if ((st = uploadMix(s, uint32_t(nch), nullptr)) != SGZ_OK) return st;
    // warm-up: the largest batch a push can produce, on the silent ring -- every lazy allocation and LDS grant of the kernels
    // happens here, not on the audio thread.  The state it leaves is cleared again.
    st = runStft(p, s->d_ring, size_t(2) * s->cap, long(s->maxFrames), s->d_mapped, nullptr, nullptr, s->stream, nullptr, /*deferLate=*/true);
    if (st == SGZ_OK) st = runDecayColour(p, s->d_mapped, long(s->maxFrames), s->d_colsBatch, s->d_linesBatch, s->d_state, s->stream);
    if (st == SGZ_OK && s->maxFrames > 1) {
        st = runStft(p, s->d_ring, size_t(2) * s->cap, 1, s->d_mapped, nullptr, nullptr, s->stream, nullptr, /*deferLate=*/true);
        if (st == SGZ_OK) st = runDecayColour(p, s->d_mapped, 1, s->d_colsBatch, s->d_linesBatch, s->d_state, s->stream);
    }
    // ... and the consumer thread's line-graph step on its own plan (one frame with line results and state)
    if (st == SGZ_OK && !isResonator(p)) st = runStft(*tp, s->d_ring, size_t(2) * s->cap, 1, s->d_lineMapped, nullptr, nullptr, s->stream);
    if (st == SGZ_OK) st = runDecayColour(*tp, isResonator(p) ? s->d_mapped : s->d_lineMapped, 1, nullptr, s->d_lines, s->d_state, s->stream);
    if (st != SGZ_OK) return st;
    if ((st = resetResonator(p, s->stream)) != SGZ_OK) return st;
    SGZ_HIP(hipMemsetAsync(s->d_state, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipMemsetAsync(s->d_lines, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipMemsetAsync(s->d_mapped, 0, size_t(s->maxFrames) * p.C * p.sides * p.P * sizeof(float), s->stream));   // (RSNT line graph: the windowed state of resonators at rest)
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

// ring position of the first of the `span` samples that end `back` samples before absolute sample count `at`
static inline uint32_t ringPos(const sgz_spectrum *s, uint64_t at, uint32_t span)
{
    return uint32_t((at % s->cap + s->cap - (span % s->cap)) % s->cap);
}

// The consumer thread's transform of the ring's newest window on its own plan.  No lock against push: read `written`, enqueue, then
// check `planned` -- an ingest kernel enqueued in front of this transform announced itself there first; if it could have reached the
// window (more than cap - W samples beyond it), the transform is simply enqueued again on the newer window.
template <typename Enqueue>
static sgz_status onNewestWindow(sgz_spectrum *s, uint32_t W, Enqueue enqueue)
{
    for (int attempt = 0; attempt < 4; ++attempt) {
        const uint64_t at = s->written.load(std::memory_order_acquire);
        const sgz_status st = enqueue(ringPos(s, at, W));
        if (st != SGZ_OK) return st;
        if (s->planned.load(std::memory_order_seq_cst) - at <= uint64_t(s->cap - W)) return SGZ_OK;
    }
    return fail(SGZ_BUSY, "the audio thread kept overtaking the render thread's window");
}

extern "C" {

sgz_status sgz_spectrum_create(const sgz_spectrum_config *cfg, sgz_spectrum **out)
{
    if (!cfg || !out) return fail(SGZ_EINVAL, "null argument");
    if (cfg->num_pairs > 16) return fail(SGZ_EINVAL, "real-time handle supports at most 32 channels");
    if (cfg->display_mode > SGZ_DISPLAY_COLOUR_SPECTRUM) return fail(SGZ_EINVAL, "display_mode: SGZ_DISPLAY_LINE_GRAPH or SGZ_DISPLAY_COLOUR_SPECTRUM");
    sgz_spectrum *s = new (std::nothrow) sgz_spectrum();
    if (!s) return fail(SGZ_ENOMEM, "out of memory");
    sgz_status st = setup(s, cfg);
    if (st != SGZ_OK) { freeHandle(s); return st; }
    *out = s;
    return SGZ_OK;
}

void sgz_spectrum_destroy(sgz_spectrum *s) { freeHandle(s); }

sgz_status sgz_spectrum_configure(sgz_spectrum *s, const sgz_spectrum_config *cfg)
{
    if (!s || !cfg) return fail(SGZ_EINVAL, "null argument");
    if (cfg->num_pairs > 16) return fail(SGZ_EINVAL, "real-time handle supports at most 32 channels");
    if (cfg->display_mode > SGZ_DISPLAY_COLOUR_SPECTRUM) return fail(SGZ_EINVAL, "display_mode: SGZ_DISPLAY_LINE_GRAPH or SGZ_DISPLAY_COLOUR_SPECTRUM");
    std::lock_guard<std::mutex> lk(s->cfgMu);
    return setup(s, cfg);
}

sgz_status sgz_spectrum_set_mix(sgz_spectrum *s, uint32_t num_sources, const uint8_t *matrix)
{
    if (!s || !matrix || num_sources == 0 || num_sources > kMaxSources) return fail(SGZ_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(s->cfgMu);
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return uploadMix(s, num_sources, matrix);
}

sgz_status sgz_spectrum_clear_state(sgz_spectrum *s)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    Plan &p = *s->plan;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    SGZ_HIP(hipMemsetAsync(s->d_state, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipMemsetAsync(s->d_lines, 0, stateN * sizeof(float), s->stream));
    return resetResonator(p, s->stream);                      // RSNT: TransformPair::clearAudioState -> cresonator.resetState (TransformPair.h:183)
}

// K_B over `frames` frames of s->d_mapped (producer's plan), the newest line results to the host's triple buffer, the columns into the queue
static sgz_status emitFrames(sgz_spectrum *s, uint32_t frames)
{
    Plan &p = *s->plan;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    sgz_status st = runDecayColour(p, s->d_mapped, long(frames), s->d_colsBatch, s->d_linesBatch, s->d_state, s->stream);
    if (st != SGZ_OK) return st;
    const float *last = s->d_linesBatch + size_t(frames - 1) * stateN;
    SGZ_HIP(hipMemcpyAsync(s->d_lines, last, stateN * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    {
        // sgz_spectrum_line_results reads the newest COMPLETED copy; `lineSeq.begun` tells it that a slot is about to be rewritten
        const uint64_t n = s->lineSeq.begin();
        const int slot = s->lineSeq.slotOf(n);
        SGZ_HIP(hipMemcpyAsync(s->h_lines + size_t(slot) * stateN, last, stateN * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        SGZ_HIP(hipEventRecord(s->lineEvents[slot], s->stream));
        s->lineSeq.publish(n);
    }
    for (uint32_t k = 0; k < frames; ++k) {
        int slot = 0;
        if (!s->colQ.producerSlot(&slot)) { s->dropped++; continue; }   // SpectrumDSP.cpp:185-186
        SGZ_HIP(hipMemcpyAsync(s->h_cols + size_t(slot) * p.P * 4, s->d_colsBatch + size_t(k) * p.P * 4, size_t(p.P) * 4,
                               hipMemcpyDeviceToHost, s->stream));
        SGZ_HIP(hipMemcpyAsync(s->d_colsQ + size_t(slot) * p.P * 4, s->d_colsBatch + size_t(k) * p.P * 4, size_t(p.P) * 4,
                               hipMemcpyDeviceToDevice, s->stream));
        SGZ_HIP(hipEventRecord(s->colEvents[slot], s->stream));
        s->colQ.producerPublish();
    }
    return SGZ_OK;
}

// one block into a staging slot and behind it the kernels that consume it; SGZ_BUSY (nothing consumed) when no slot is free
static sgz_status spectrumPushNow(sgz_spectrum *s, const float *const *blk, uint32_t nch, uint32_t n)
{
    Plan &p = *s->plan;
    TraceRange range("sgz::spectrum push (stage + ingest + frames)");
    const uint32_t pieces = (n + kPiece - 1) / kPiece;
    // all or nothing: every piece's staging slot must be free now
    for (uint32_t k = 0; k < pieces; ++k) {
        const int slot = int((s->stage.seq + k) % StageRing::kSlots);
        if (s->stage.used[slot] && hipEventQuery(s->stage.ev[slot]) == hipErrorNotReady) return SGZ_BUSY;
    }
    const uint32_t numDst = 2 * p.C, W = p.W, hop = p.cfg.hop;
    const bool lineGraph = p.cfg.display_mode == SGZ_DISPLAY_LINE_GRAPH;
    const bool strict = s->strictQuirks && !lineGraph && !isResonator(p);     // (RSNT: resonatingDispatch gets `buffer + offset`, :1180 -- its frames are the ideal ones)
    // strict mode: the callback's frames up front.  audioEntryPoint's loop (TransformDSP.inl:1170-1204) fires its first frame after
    // `hop - processedSamplesSinceLastFrame` samples and then one per `hop`; frame j is prepared from `availableSamples` of the UN-offset
    // block: stop_0 = min(first, W), stop_j = min(hop, W) afterwards
    const uint64_t blockStart = s->written.load(std::memory_order_relaxed);
    uint32_t strictFrames = 0, strictFirst = 0, strictDone = 0;
    if (strict) {
        strictFirst = s->sinceLast > hop ? 0u : hop - s->sinceLast;
        if (n >= strictFirst && n > 0) strictFrames = (n - strictFirst) / hop + 1;
    }
    const uint32_t extra = s->audioHistory > W ? uint32_t(s->audioHistory - W) : 0u;
    const float *ptrs[kMaxSources];
    for (uint32_t done = 0; done < n;) {
        const uint32_t m = std::min(n - done, kPiece);
        for (uint32_t c = 0; c < nch; ++c) ptrs[c] = blk[c] + done;
        sgz_status st;
        const float *d_block = s->stage.stage(ptrs, m, s->stream, &st);
        if (!d_block) return st;
        const uint64_t at = s->written.load(std::memory_order_relaxed);
        s->planned.store(at + m, std::memory_order_seq_cst);          // (before the launch: see the struct)
        hipLaunchKernelGGL(ringIngestKernel, dim3((m + 255) / 256, numDst), dim3(256), 0, s->stream, d_block, m, nch, s->d_mix,
                           s->d_ring, s->cap, numDst, uint32_t(at % s->cap));
        SGZ_HIP(hipGetLastError());
        if ((st = s->stage.commit(s->stream)) != SGZ_OK) return st;
        if (lineGraph) {
            // TransformDSP.inl:1167 / :1206-1209: no transform on the audio thread; the resonators take the whole block
            if (isResonator(p))
                if ((st = runResonatorAdvance(p, s->d_ring + uint32_t(at % s->cap), size_t(2) * s->cap, m, s->d_mapped, s->stream)) != SGZ_OK) return st;
        } else if (strict) {
            // the frames whose `stop` samples are in the ring now (all of them after the first piece unless hop > 16384)
            while (strictDone < strictFrames) {
                uint32_t batch = 0;
                while (strictDone + batch < strictFrames && batch < s->maxFrames) {
                    const uint32_t j = strictDone + batch;
                    const uint32_t stop = std::min(j == 0 ? strictFirst : hop, W);
                    if (stop > done + m) break;
                    float *out = s->d_mapped + size_t(batch) * p.C * p.sides * p.P;
                    if (extra && stop + extra <= W) {                                           // Q2: a shortened, shifted frame
                        hipLaunchKernelGGL(strictGatherKernel, dim3((W + 255) / 256, numDst), dim3(256), 0, s->stream, s->d_ring, s->cap,
                                           ringPos(s, blockStart + stop, W), W, stop, extra, s->d_strict);
                        SGZ_HIP(hipGetLastError());
                        st = runStft(p, s->d_strict, size_t(W), 1, out, nullptr, nullptr, s->stream);
                    } else {                                                                    // Q1: the window that ends `stop` samples into the block
                        st = runStft(p, s->d_ring + ringPos(s, blockStart + stop, W), size_t(2) * s->cap, 1, out, nullptr, nullptr, s->stream);
                    }
                    if (st != SGZ_OK) return st;
                    ++batch;
                }
                if (!batch) break;
                if ((st = emitFrames(s, batch)) != SGZ_OK) return st;
                strictDone += batch;
            }
        } else {
            // frames that fire inside this piece (TransformDSP.inl:1172-1185): the first after hop - sinceLast samples, then every hop
            const uint32_t first = s->sinceLast >= hop ? 0u : hop - s->sinceLast;
            uint32_t frames = 0;
            if (first <= m && (first > 0 || s->sinceLast >= hop)) frames = (m - first) / hop + 1;
            if (frames) {
                // frame k's window ends `first + k hop` samples into the piece; in the mirrored ring it starts at q + k hop, contiguous
                // (RSNT: the frames' hop-sample segments tile the stream -- frame k consumes [end_k - hop, end_k))
                const uint32_t q = ringPos(s, at + first, isResonator(p) ? hop : W);
                st = runStft(p, s->d_ring + q, size_t(2) * s->cap, long(frames), s->d_mapped, nullptr, nullptr, s->stream, nullptr, /*deferLate=*/true);
                if (st != SGZ_OK) return st;
                if ((st = emitFrames(s, frames)) != SGZ_OK) return st;
                s->sinceLast = (m - first) - (frames - 1) * hop;
            } else s->sinceLast += m;
        }
        s->written.store(at + m, std::memory_order_release);
        done += m;
    }
    if (strict) s->sinceLast = strictFrames ? (n - strictFirst) % hop : s->sinceLast + n;
    return SGZ_OK;
}

sgz_status sgz_spectrum_push(sgz_spectrum *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples)
{
    if (!s || !planar) return fail(SGZ_EINVAL, "null argument");
    std::unique_lock<std::mutex> lk(s->cfgMu, std::try_to_lock);     // never waits: a reconfiguration in progress refuses the block
    if (!lk.owns_lock()) { s->busy++; return SGZ_BUSY; }
    Plan &p = *s->plan;
    if (num_channels != s->numSources)
        return fail(SGZ_EINVAL, "num_channels must equal 2*num_pairs (SpectrumDSP.cpp:65-72), or the source count of sgz_spectrum_set_mix");
    if ((nsamples + kPiece - 1) / kPiece > uint32_t(StageRing::kSlots)) return fail(SGZ_EINVAL, "push takes at most 131072 samples per call");
    auto pushNow = [&](const float *const *blk, uint32_t nch, uint32_t n) -> sgz_status { return spectrumPushNow(s, blk, nch, n); };
    // never waits: a block the GPU is not ready for queues up behind the earlier ones (rt_common.hpp Backlog); SGZ_BUSY = that FIFO is full
    const sgz_status st = pushThroughBacklog(s->backlog, planar, num_channels, nsamples, pushNow);
    if (st == SGZ_BUSY) s->busy++;
    s->deferredStat.store(s->backlog.deferred, std::memory_order_relaxed);
    s->waitingStat.store(s->backlog.count, std::memory_order_relaxed);
    return st;
}

sgz_status sgz_spectrum_flush(sgz_spectrum *s)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->cfgMu);
    const float *ptrs[64];
    while (s->backlog.count) {
        const Backlog::Entry e = s->backlog.front();
        for (uint32_t c = 0; c < e.channels && c < 64; ++c) ptrs[c] = s->backlog.buf + e.off + size_t(c) * e.n;
        const sgz_status st = spectrumPushNow(s, ptrs, e.channels, e.n);
        if (st == SGZ_BUSY) { SGZ_HIP(hipStreamSynchronize(s->stream)); continue; }      // this call may wait: it is not the audio thread's
        s->backlog.pop();
        s->waitingStat.store(s->backlog.count, std::memory_order_relaxed);
        if (st != SGZ_OK) return st;
    }
    return SGZ_OK;
}

sgz_status sgz_spectrum_pop_column(sgz_spectrum *s, uint8_t *rgba, uint32_t *axis_points)
{
    if (!s || !rgba) return fail(SGZ_EINVAL, "null argument");
    // a LINE_GRAPH handle (display_mode 0: what a zero-initialised configuration asks for, as in the reference's enum) produces no columns:
    // say so instead of staying empty for ever
    if (s->plan->cfg.display_mode == SGZ_DISPLAY_LINE_GRAPH) return fail(SGZ_EINVAL, "a LINE_GRAPH handle has no colour columns (sgz_spectrum_config::display_mode)");
    const uint64_t head = s->colQ.consumerHead();
    if (!s->colQ.consumerHas(head)) return SGZ_EMPTY;
    const int slot = int(head % kQueueDepth);
    const hipError_t q = hipEventQuery(s->colEvents[slot]);
    if (q == hipErrorNotReady) return SGZ_EMPTY;
    if (q != hipSuccess) return hipFail(q, "hipEventQuery");
    const Plan &p = *s->plan;
    std::memcpy(rgba, s->h_cols + size_t(slot) * p.P * 4, size_t(p.P) * 4);
    if (axis_points) *axis_points = p.P;
    s->colQ.consumerRelease(head + 1);
    return SGZ_OK;
}

sgz_status sgz_spectrum_bind_image(sgz_spectrum *s, void *d_image, uint32_t columns, size_t pitch_bytes)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    unbindImage(s);
    if (!d_image) return SGZ_OK;
    if (columns == 0 || pitch_bytes < size_t(columns) * 4 || (pitch_bytes & 3) || (reinterpret_cast<uintptr_t>(d_image) & 3))
        return fail(SGZ_EINVAL, "image: columns > 0, pitch >= 4 * columns, 4-byte aligned");
    s->d_image = static_cast<uint8_t *>(d_image); s->imgColumns = columns; s->imgPitch = pitch_bytes; s->imgX = 0;
    return SGZ_OK;
}

sgz_status sgz_spectrum_create_image(sgz_spectrum *s, uint32_t columns, void **d_image, size_t *pitch_bytes, int *dmabuf_fd)
{
    if (!s || columns == 0 || !pitch_bytes) return fail(SGZ_EINVAL, "bad argument");
    unbindImage(s);
    const Plan &p = *s->plan;
    const size_t pitch = (size_t(columns) * 4 + 255) & ~size_t(255);
    // a dma-buf is a whole buffer object: allocated in whole 2 MiB blocks so that the fd describes the image and nothing else (api.hip
    // sgz_export_alloc has the measurement)
    const size_t bytes = (pitch * p.P + kExportGranule - 1) & ~(kExportGranule - 1);
    uint8_t *img = nullptr;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&img), bytes));
    hipError_t e = hipMemset(img, 0, bytes);
    int fd = -1;
    if (e == hipSuccess && dmabuf_fd) e = hipMemGetHandleForAddressRange(&fd, img, bytes, hipMemRangeHandleTypeDmaBufFd, 0);
    if (e != hipSuccess) { (void)hipFree(img); return hipFail(e, "image allocation / dma-buf export"); }
    s->d_image = img; s->imgOwned = true; s->imgColumns = columns; s->imgPitch = pitch; s->imgX = 0;
    if (d_image) *d_image = img;
    *pitch_bytes = pitch;
    if (dmabuf_fd) *dmabuf_fd = fd;
    return SGZ_OK;
}

sgz_status sgz_spectrum_bind_gl_buffer(sgz_spectrum *s, unsigned int gl_buffer, uint32_t columns, size_t pitch_bytes)
{
    if (!s || columns == 0 || pitch_bytes < size_t(columns) * 4 || (pitch_bytes & 3)) return fail(SGZ_EINVAL, "bad argument");
    unbindImage(s);
    const Plan &p = *s->plan;
    hipGraphicsResource *res = nullptr;
    hipError_t e = hipGraphicsGLRegisterBuffer(&res, gl_buffer, hipGraphicsRegisterFlagsWriteDiscard);
    if (e != hipSuccess || !res) return hipFail(e != hipSuccess ? e : hipErrorInvalidValue, "hipGraphicsGLRegisterBuffer (needs a current OpenGL context on this device)");
    void *ptr = nullptr; size_t size = 0;
    e = hipGraphicsMapResources(1, &res, s->outStream);
    if (e == hipSuccess) e = hipGraphicsResourceGetMappedPointer(&ptr, &size, res);
    // (mapped here only to check its size: GL may touch the buffer whenever HIP does not hold it mapped, so flush_columns maps and
    // unmaps it around its own writes -- the interop contract)
    const hipError_t eu = hipGraphicsUnmapResources(1, &res, s->outStream);
    if (e != hipSuccess || eu != hipSuccess || size < pitch_bytes * p.P) {
        (void)hipGraphicsUnregisterResource(res);
        return e != hipSuccess ? hipFail(e, "mapping the GL buffer") : eu != hipSuccess ? hipFail(eu, "unmapping the GL buffer")
                                                                     : fail(SGZ_EINVAL, "GL buffer smaller than pitch * axis_points");
    }
    s->glResource = res;
    s->d_image = nullptr; s->imgColumns = columns; s->imgPitch = pitch_bytes; s->imgX = 0;
    return SGZ_OK;
}

sgz_status sgz_spectrum_flush_columns(sgz_spectrum *s, uint32_t *first_column, uint32_t *count)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    if (s->plan->cfg.display_mode == SGZ_DISPLAY_LINE_GRAPH) return fail(SGZ_EINVAL, "a LINE_GRAPH handle has no colour columns (sgz_spectrum_config::display_mode)");
    if (!s->d_image && !s->glResource) return fail(SGZ_EINVAL, "no image bound");
    const Plan &p = *s->plan;
    uint8_t *image = s->d_image;
    hipGraphicsResource *gl = static_cast<hipGraphicsResource *>(s->glResource);
    if (gl) {                                                  // a GL buffer is HIP's only between map and unmap
        void *ptr = nullptr; size_t size = 0;
        SGZ_HIP(hipGraphicsMapResources(1, &gl, s->outStream));
        const hipError_t e = hipGraphicsResourceGetMappedPointer(&ptr, &size, gl);
        if (e != hipSuccess) { (void)hipGraphicsUnmapResources(1, &gl, s->outStream); return hipFail(e, "hipGraphicsResourceGetMappedPointer"); }
        image = static_cast<uint8_t *>(ptr);
    }
    uint32_t n = 0;
    const uint32_t first = s->imgX;
    uint64_t head = s->colQ.consumerHead();
    // at most one lap of the image per call, so that `first` / `count` describe the dirty range unambiguously
    while (n < s->imgColumns && s->colQ.consumerHas(head)) {
        const int slot = int(head % kQueueDepth);
        const hipError_t q = hipEventQuery(s->colEvents[slot]);
        if (q == hipErrorNotReady) break;
        hipError_t e = q;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(columnScatterKernel, dim3((p.P + 255) / 256), dim3(256), 0, s->outStream,
                               reinterpret_cast<const uint32_t *>(s->d_colsQ + size_t(slot) * p.P * 4), image, s->imgPitch, s->imgX, p.P);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {                                   // (a GL buffer never stays mapped behind an error return)
            if (gl) { (void)hipStreamSynchronize(s->outStream); (void)hipGraphicsUnmapResources(1, &gl, s->outStream); }
            return hipFail(e, "sgz_spectrum_flush_columns");
        }
        s->imgX = (s->imgX + 1) % s->imgColumns;           // framePixelPosition %= numSpectrumColumns (SpectrumRendering.cpp:712-718)
        ++head; ++n;
    }
    if (n) {
        const hipError_t e = hipStreamSynchronize(s->outStream);     // the texels are in place before the slots go back to the producer
        if (e != hipSuccess) {
            if (gl) (void)hipGraphicsUnmapResources(1, &gl, s->outStream);
            return hipFail(e, "hipStreamSynchronize");
        }
        s->colQ.consumerRelease(head);
    }
    if (gl) SGZ_HIP(hipGraphicsUnmapResources(1, &gl, s->outStream));
    if (first_column) *first_column = first;
    if (count) *count = n;
    return n ? SGZ_OK : SGZ_EMPTY;
}

sgz_status sgz_spectrum_line_results(sgz_spectrum *s, uint32_t pair, uint32_t graph, float *out)
{
    if (!s || !out) return fail(SGZ_EINVAL, "null argument");
    const Plan &p = *s->plan;
    if (pair >= p.C || graph >= SGZ_NUM_GRAPHS) return fail(SGZ_EINVAL, "pair/graph out of range");
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2, at = (size_t(pair) * SGZ_NUM_GRAPHS + graph) * p.P * 2, bytes = size_t(p.P) * 2 * sizeof(float);
    if (p.cfg.display_mode == SGZ_DISPLAY_LINE_GRAPH) {         // the consumer's own results (sgz_spectrum_render_lines)
        std::memcpy(out, s->h_lineOut + at, bytes);
        return SGZ_OK;
    }
    // the newest copy that has arrived: nothing is waited for and the producer's stream is not touched.  Seqlock: a slot is rewritten
    // by copy number n + kLineSlots, which `lineSeq.begun` announces first -- a read that may have overlapped it is repeated on a newer slot.
    auto landed = [&](int slot) { return hipEventQuery(s->lineEvents[slot]) == hipSuccess; };
    for (int attempt = 0; attempt < 8; ++attempt) {
        bool none = false;
        const uint64_t n = s->lineSeq.newest(landed, &none);
        if (n == 0) {
            if (none) { std::memset(out, 0, bytes); return SGZ_OK; }                    // no frame yet (lineGraphs start zeroed)
            continue;                                               // (every candidate still in flight: look again)
        }
        std::memcpy(out, s->h_lines + size_t(s->lineSeq.slotOf(n)) * stateN + at, bytes);
        if (s->lineSeq.stillValid(n)) return SGZ_OK;
    }
    // the producer lapped the reader eight times in a row: fall back to the device copy behind the producer's work (waits)
    SGZ_HIP(hipMemcpyAsync(out, s->d_lines + at, bytes, hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

sgz_status sgz_spectrum_render_lines(sgz_spectrum *s, const float *poles, float *out)
{
    if (!s || !out) return fail(SGZ_EINVAL, "null argument");
    Plan &p = *s->trackPlan;                                  // the consumer's plan (per-launch scratch is per plan)
    TraceRange range("sgz::spectrum render_lines");
    if (p.cfg.display_mode != SGZ_DISPLAY_LINE_GRAPH) return fail(SGZ_EINVAL, "sgz_spectrum_render_lines: the handle is configured for SGZ_DISPLAY_COLOUR_SPECTRUM");
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    const float *mapped = s->d_lineMapped;
    sgz_status st = SGZ_OK;
    if (isResonator(p)) {
        // mapToLinearSpace's RSNT branch (:1103-1133): the windowed state as of the last pushed block -- the producer's advance leaves
        // it in d_mapped (stream order makes this read see a whole block's result)
        mapped = s->d_mapped;
    } else {
        st = onNewestWindow(s, p.W, [&](uint32_t q) { return runStft(p, s->d_ring + q, size_t(2) * s->cap, 1, s->d_lineMapped, nullptr, nullptr, s->stream); });
        if (st != SGZ_OK) return st;
    }
    // postProcessStdTransform (:1438): both graphs' filters advance ONCE per rendered frame, with this frame's poles
    const DeviceScalars keep = p.scalars;
    if (poles)
        for (int k = 0; k < SGZ_NUM_GRAPHS; ++k) { p.scalars.pole[k] = poles[k]; p.scalars.phasePole[k] = std::pow(poles[k], 0.3f); }
    st = runDecayColour(p, mapped, 1, nullptr, s->d_lines, s->d_state, s->stream);
    p.scalars = keep;
    if (st != SGZ_OK) return st;
    SGZ_HIP(hipMemcpyAsync(s->h_lineOut, s->d_lines, stateN * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    std::memcpy(out, s->h_lineOut, stateN * sizeof(float));
    return SGZ_OK;
}

sgz_status sgz_spectrum_set_option(sgz_spectrum *s, uint32_t option, uint64_t value)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->cfgMu);
    switch (option) {
    case SGZ_RT_OPT_STRICT_REFERENCE_QUIRKS: s->strictQuirks = value != 0; return SGZ_OK;
    case SGZ_RT_OPT_AUDIO_HISTORY:
        if (value != 0 && value < s->plan->W) return fail(SGZ_EINVAL, "the audio history holds at least one window (prepareTransform refuses shorter ones, TransformDSP.inl:241-242)");
        s->audioHistory = value;
        return SGZ_OK;
    default: return fail(SGZ_EINVAL, "unknown handle option");
    }
}

void *sgz_spectrum_stream(sgz_spectrum *s) { return s ? s->stream : nullptr; }

sgz_status sgz_spectrum_backlog(sgz_spectrum *s, uint64_t *deferred_blocks, uint32_t *waiting_now)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    // (the FIFO belongs to the producer, which mirrors its counters into atomics: no lock, so that a UI thread polling this can never
    // make a push find the handle held)
    if (deferred_blocks) *deferred_blocks = s->deferredStat.load(std::memory_order_relaxed);
    if (waiting_now) *waiting_now = s->waitingStat.load(std::memory_order_relaxed);
    return SGZ_OK;
}

sgz_status sgz_spectrum_stats(sgz_spectrum *s, uint64_t *dropped_columns, uint64_t *refused_pushes)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    if (dropped_columns) *dropped_columns = s->dropped.load();
    if (refused_pushes) *refused_pushes = s->busy.load();
    return SGZ_OK;
}

sgz_status sgz_spectrum_track_peak(sgz_spectrum *s, uint32_t pair, double mouse_fraction, sgz_peak *out)
{
    if (!s || !out) return fail(SGZ_EINVAL, "null argument");
    Plan &p = *s->trackPlan;                                  // (not the producer's plan: see sgz_spectrum::trackPlan)
    if (pair >= p.C) return fail(SGZ_EINVAL, "pair out of range");
    if (!s->d_trackBins) return fail(SGZ_EUNSUPPORTED, "frequency tracker: magnitude modes of the FFT algorithm only");
    // the window a frame firing now would transform (work already enqueued by push precedes this on the stream)
    sgz_status st = onNewestWindow(s, p.W, [&](uint32_t q) { return runStft(p, s->d_ring + q, size_t(2) * s->cap, 1, nullptr, s->d_trackBins, nullptr, s->stream); });
    if (st != SGZ_OK) return st;
    if ((st = runTrackPeak(p, s->d_trackBins + size_t(pair) * (size_t(p.N) + 1), mouse_fraction, s->d_peak, s->stream)) != SGZ_OK) return st;
    SGZ_HIP(hipMemcpyAsync(out, s->d_peak, sizeof(sgz_peak), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

sgz_status sgz_spectrum_track_peak_lines(sgz_spectrum *s, uint32_t pair, uint32_t graph, double mouse_fraction, sgz_line_peak *out)
{
    if (!s || !out) return fail(SGZ_EINVAL, "null argument");
    const Plan &p = *s->plan;
    std::vector<float> results(size_t(p.P) * 2);
    if (sgz_status st = sgz_spectrum_line_results(s, pair, graph, results.data()); st != SGZ_OK) return st;
    return trackPeakLines(p, results.data(), mouse_fraction, out);
}

/* parity hook: the W newest samples of destination channel `channel` as K_A would read them (one contiguous range of the mirrored
 * ring) */
sgz_status sgz_spectrum_history(sgz_spectrum *s, uint32_t channel, float *out)
{
    if (!s || !out || channel >= 2 * s->plan->C) return fail(SGZ_EINVAL, "bad argument");
    const uint32_t W = s->plan->W;
    const uint32_t q = ringPos(s, s->written.load(std::memory_order_acquire), W);
    SGZ_HIP(hipMemcpyAsync(out, s->d_ring + size_t(channel) * 2 * s->cap + q, size_t(W) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

}  // extern "C"
