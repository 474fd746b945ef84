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
// Framing is the ideal STFT framing (a frame fires every `hop` samples and covers the W samples that end at the firing point):
// the reference's within-callback offset quirk is deliberately not reproduced.  All frames that fire inside one
// staged piece go through K_A and K_B as ONE launch each (their windows are hop-spaced ranges of the same ring).
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
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "rt_common.hpp"

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
    std::atomic<uint32_t> head{0};    // write position (mod cap): written by the producer, read by the tracker on the consumer thread
    uint32_t sinceLast = 0;           // processedSamplesSinceLastFrame
    uint32_t maxFrames = 1;
    uint8_t *d_mix = nullptr;
    uint32_t numSources = 0;
    float *d_mapped = nullptr, *d_state = nullptr, *d_lines = nullptr, *d_linesBatch = nullptr;
    float *d_trackBins = nullptr; sgz_peak *d_peak = nullptr;     // frequency tracker: csf of the newest window [C][N + 1], its result
    uint8_t *d_colsBatch = nullptr;   // [maxFrames][P][4]
    uint8_t *h_cols = nullptr;        // pinned [kQueueDepth][P][4]
    hipEvent_t colEvents[kQueueDepth] = {};
    // SPSC column queue: the producer fills slot tail % depth and bumps tail, the consumer reads slot head % depth and bumps head
    std::atomic<uint64_t> qHead{0}, qTail{0};
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
    for (float *p : {s->d_ring, s->d_mapped, s->d_state, s->d_lines, s->d_linesBatch, s->d_trackBins}) if (p) (void)hipFree(p);
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
    if (sgz_status st = s->backlog.init(size_t(numSources) * std::max<size_t>(size_t(s->plan->cfg.sample_rate), size_t(32) * kPiece)); st != SGZ_OK) return st;
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
    for (float **q : {&s->d_ring, &s->d_mapped, &s->d_state, &s->d_lines, &s->d_linesBatch, &s->d_trackBins}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    if (s->d_colsBatch) { (void)hipFree(s->d_colsBatch); s->d_colsBatch = nullptr; }
    if (s->h_cols) { (void)hipHostFree(s->h_cols); s->h_cols = nullptr; }
    if (s->d_colsQ) { (void)hipFree(s->d_colsQ); s->d_colsQ = nullptr; }
    unbindImage(s);                                        // the image's height is the axis size: a new configuration needs a new binding
    // a piece's frames read windows that end inside the piece: the ring must hold W + one piece (RSNT: a frame consumes the `hop`
    // samples that end with it)
    s->cap = ((isResonator(p) ? p.cfg.hop : p.W) + kPiece + 63u) & ~63u;
    s->maxFrames = kPiece / p.cfg.hop + 1;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_ring), nch * 2 * s->cap * sizeof(float)));
    SGZ_HIP(hipMemsetAsync(s->d_ring, 0, nch * 2 * s->cap * sizeof(float), s->stream));    // history starts as silence
    s->head.store(0);
    s->sinceLast = 0;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_mapped), size_t(s->maxFrames) * p.C * p.sides * p.P * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_state), stateN * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_lines), stateN * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_linesBatch), size_t(s->maxFrames) * stateN * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_colsBatch), size_t(s->maxFrames) * p.P * 4));
    if (p.cfg.channel_mode != SGZ_CH_PHASE && !isResonator(p)) SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_trackBins), size_t(p.C) * (size_t(p.N) + 1) * sizeof(float)));
    if (!s->d_peak) SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_peak), sizeof(sgz_peak)));
    SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_cols), size_t(kQueueDepth) * p.P * 4, hipHostMallocDefault));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_colsQ), size_t(kQueueDepth) * p.P * 4));
    if (!s->outStream) SGZ_HIP(hipStreamCreateWithFlags(&s->outStream, hipStreamNonBlocking));
    for (auto &e : s->colEvents) if (!e) SGZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->qHead.store(0); s->qTail.store(0);
    if ((st = uploadMix(s, uint32_t(nch), nullptr)) != SGZ_OK) return st;
    // warm-up: the largest batch a push can produce, on the silent ring -- every lazy allocation and LDS grant of the kernels
    // happens here, not on the audio thread.  The state it leaves is cleared again.
    st = runStft(p, s->d_ring, size_t(2) * s->cap, long(s->maxFrames), s->d_mapped, nullptr, nullptr, s->stream, nullptr, /*deferLate=*/true);
    if (st == SGZ_OK) st = runDecayColour(p, s->d_mapped, long(s->maxFrames), s->d_colsBatch, s->d_linesBatch, s->d_state, s->stream);
    if (st == SGZ_OK && s->maxFrames > 1) {
        st = runStft(p, s->d_ring, size_t(2) * s->cap, 1, s->d_mapped, nullptr, nullptr, s->stream, nullptr, /*deferLate=*/true);
        if (st == SGZ_OK) st = runDecayColour(p, s->d_mapped, 1, s->d_colsBatch, s->d_linesBatch, s->d_state, s->stream);
    }
    if (st != SGZ_OK) return st;
    if ((st = resetResonator(p, s->stream)) != SGZ_OK) return st;
    SGZ_HIP(hipMemsetAsync(s->d_state, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipMemsetAsync(s->d_lines, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

extern "C" {

sgz_status sgz_spectrum_create(const sgz_spectrum_config *cfg, sgz_spectrum **out)
{
    if (!cfg || !out) return fail(SGZ_EINVAL, "null argument");
    if (cfg->num_pairs > 16) return fail(SGZ_EINVAL, "real-time handle supports at most 32 channels");
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

// one block into a staging slot and behind it the kernels that consume it; SGZ_BUSY (nothing consumed) when no slot is free
static sgz_status spectrumPushNow(sgz_spectrum *s, const float *const *blk, uint32_t nch, uint32_t n)
{
    Plan &p = *s->plan;
    const uint32_t pieces = (n + kPiece - 1) / kPiece;
    // all or nothing: every piece's staging slot must be free now
    for (uint32_t k = 0; k < pieces; ++k) {
        const int slot = int((s->stage.seq + k) % StageRing::kSlots);
        if (s->stage.used[slot] && hipEventQuery(s->stage.ev[slot]) == hipErrorNotReady) return SGZ_BUSY;
    }
    const uint32_t numDst = 2 * p.C, W = p.W, hop = p.cfg.hop;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    const float *ptrs[kMaxSources];
    for (uint32_t done = 0; done < n;) {
        const uint32_t m = std::min(n - done, kPiece);
        for (uint32_t c = 0; c < nch; ++c) ptrs[c] = blk[c] + done;
        sgz_status st;
        const float *d_block = s->stage.stage(ptrs, m, s->stream, &st);
        if (!d_block) return st;
        hipLaunchKernelGGL(ringIngestKernel, dim3((m + 255) / 256, numDst), dim3(256), 0, s->stream, d_block, m, nch, s->d_mix,
                           s->d_ring, s->cap, numDst, s->head.load(std::memory_order_relaxed));
        SGZ_HIP(hipGetLastError());
        if ((st = s->stage.commit(s->stream)) != SGZ_OK) return st;
        // frames that fire inside this piece (TransformDSP.inl:1172-1185): the first after hop - sinceLast samples, then every hop
        const uint32_t first = s->sinceLast >= hop ? 0u : hop - s->sinceLast;
        uint32_t frames = 0;
        if (first <= m && (first > 0 || s->sinceLast >= hop)) frames = (m - first) / hop + 1;
        if (frames) {
            // frame k's window ends `first + k hop` samples into the piece; in the mirrored ring it starts at q + k hop, contiguous
            const uint32_t end0 = (s->head.load(std::memory_order_relaxed) + first) % s->cap;
            // (RSNT: the frames' hop-sample segments tile the stream -- frame k consumes [end_k - hop, end_k))
            const uint32_t span = isResonator(p) ? hop : W;
            const uint32_t q = (end0 + s->cap - (span % s->cap)) % s->cap;
            st = runStft(p, s->d_ring + q, size_t(2) * s->cap, long(frames), s->d_mapped, nullptr, nullptr, s->stream, nullptr, /*deferLate=*/true);
            if (st != SGZ_OK) return st;
            st = runDecayColour(p, s->d_mapped, long(frames), s->d_colsBatch, s->d_linesBatch, s->d_state, s->stream);
            if (st != SGZ_OK) return st;
            SGZ_HIP(hipMemcpyAsync(s->d_lines, s->d_linesBatch + size_t(frames - 1) * stateN, stateN * sizeof(float), hipMemcpyDeviceToDevice,
                                   s->stream));
            for (uint32_t k = 0; k < frames; ++k) {
                const uint64_t tail = s->qTail.load(std::memory_order_relaxed);
                if (tail - s->qHead.load(std::memory_order_acquire) >= uint64_t(kQueueDepth)) { s->dropped++; continue; }   // SpectrumDSP.cpp:185-186
                const int slot = int(tail % kQueueDepth);
                SGZ_HIP(hipMemcpyAsync(s->h_cols + size_t(slot) * p.P * 4, s->d_colsBatch + size_t(k) * p.P * 4, size_t(p.P) * 4,
                                       hipMemcpyDeviceToHost, s->stream));
                SGZ_HIP(hipMemcpyAsync(s->d_colsQ + size_t(slot) * p.P * 4, s->d_colsBatch + size_t(k) * p.P * 4, size_t(p.P) * 4,
                                       hipMemcpyDeviceToDevice, s->stream));
                SGZ_HIP(hipEventRecord(s->colEvents[slot], s->stream));
                s->qTail.store(tail + 1, std::memory_order_release);
            }
            s->sinceLast = (m - first) - (frames - 1) * hop;
        } else s->sinceLast += m;
        s->head.store((s->head.load(std::memory_order_relaxed) + m) % s->cap, std::memory_order_release);
        done += m;
    }
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
        if (st != SGZ_OK) return st;
    }
    return SGZ_OK;
}

sgz_status sgz_spectrum_pop_column(sgz_spectrum *s, uint8_t *rgba, uint32_t *axis_points)
{
    if (!s || !rgba) return fail(SGZ_EINVAL, "null argument");
    const uint64_t head = s->qHead.load(std::memory_order_relaxed);
    if (head == s->qTail.load(std::memory_order_acquire)) return SGZ_EMPTY;
    const int slot = int(head % kQueueDepth);
    const hipError_t q = hipEventQuery(s->colEvents[slot]);
    if (q == hipErrorNotReady) return SGZ_EMPTY;
    if (q != hipSuccess) return hipFail(q, "hipEventQuery");
    const Plan &p = *s->plan;
    std::memcpy(rgba, s->h_cols + size_t(slot) * p.P * 4, size_t(p.P) * 4);
    if (axis_points) *axis_points = p.P;
    s->qHead.store(head + 1, std::memory_order_release);
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
    uint64_t head = s->qHead.load(std::memory_order_relaxed);
    // at most one lap of the image per call, so that `first` / `count` describe the dirty range unambiguously
    while (n < s->imgColumns && head != s->qTail.load(std::memory_order_acquire)) {
        const int slot = int(head % kQueueDepth);
        const hipError_t q = hipEventQuery(s->colEvents[slot]);
        if (q == hipErrorNotReady) break;
        if (q != hipSuccess) return hipFail(q, "hipEventQuery");
        hipLaunchKernelGGL(columnScatterKernel, dim3((p.P + 255) / 256), dim3(256), 0, s->outStream,
                           reinterpret_cast<const uint32_t *>(s->d_colsQ + size_t(slot) * p.P * 4), image, s->imgPitch, s->imgX, p.P);
        SGZ_HIP(hipGetLastError());
        s->imgX = (s->imgX + 1) % s->imgColumns;           // framePixelPosition %= numSpectrumColumns (SpectrumRendering.cpp:712-718)
        ++head; ++n;
    }
    if (n) {
        SGZ_HIP(hipStreamSynchronize(s->outStream));        // the texels are in place before the slots go back to the producer
        s->qHead.store(head, std::memory_order_release);
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
    const float *src = s->d_lines + (size_t(pair) * SGZ_NUM_GRAPHS + graph) * p.P * 2;
    SGZ_HIP(hipMemcpyAsync(out, src, size_t(p.P) * 2 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

void *sgz_spectrum_stream(sgz_spectrum *s) { return s ? s->stream : nullptr; }

sgz_status sgz_spectrum_backlog(sgz_spectrum *s, uint64_t *deferred_blocks, uint32_t *waiting_now)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->cfgMu);                 // (the FIFO belongs to the producer: looked at under the push lock)
    if (deferred_blocks) *deferred_blocks = s->backlog.deferred;
    if (waiting_now) *waiting_now = s->backlog.count;
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
    const uint32_t q = (s->head.load(std::memory_order_acquire) + s->cap - (p.W % s->cap)) % s->cap;
    sgz_status st = runStft(p, s->d_ring + q, size_t(2) * s->cap, 1, nullptr, s->d_trackBins, nullptr, s->stream);
    if (st != SGZ_OK) return st;
    if ((st = runTrackPeak(p, s->d_trackBins + size_t(pair) * (size_t(p.N) + 1), mouse_fraction, s->d_peak, s->stream)) != SGZ_OK) return st;
    SGZ_HIP(hipMemcpyAsync(out, s->d_peak, sizeof(sgz_peak), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

/* parity hook: the W newest samples of destination channel `channel` as K_A would read them (one contiguous range of the mirrored
 * ring) */
sgz_status sgz_spectrum_history(sgz_spectrum *s, uint32_t channel, float *out)
{
    if (!s || !out || channel >= 2 * s->plan->C) return fail(SGZ_EINVAL, "bad argument");
    const uint32_t W = s->plan->W;
    const uint32_t q = (s->head.load(std::memory_order_acquire) + s->cap - (W % s->cap)) % s->cap;
    SGZ_HIP(hipMemcpyAsync(out, s->d_ring + size_t(channel) * 2 * s->cap + q, size_t(W) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

}  // extern "C"
