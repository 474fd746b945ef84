// realtime.hip -- the per-block real-time Spectrum path behind sgz_spectrum_push / sgz_spectrum_pop_column.
//
// Replaces Spectrum::ProcessorShell::onStreamAudio -> AudioDispatcher::dispatch
// (Source/Spectrum/SpectrumDSP.cpp:63-108, :210-216), TransformPair::audioEntryPoint's frame cadence
// (TransformDSP.inl:1165-1211: processedSamplesSinceLastFrame / sampleBufferSize) and the consumer side of
// the frameQueue (SpectrumRendering.cpp:696-721).  The audio history (cpl::AudioStream's circular
// buffer in the reference) is mirrored in HBM as a planar linear buffer per channel; frames read their W
// newest samples straight from it.  Framing is the ideal STFT framing (a frame fires every `hop` samples
// and covers the W samples that end at the firing point): the reference's within-callback offset quirk
//  is deliberately not reproduced.
//
// Threading: one producer thread (push) and one consumer thread (pop_column / line_results); push only
// enqueues work on the handle's stream and returns.
#include <hip/hip_runtime.h>

#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <vector>

#include "runtime.hpp"

using namespace sgz;

namespace {
constexpr int kQueueDepth = 10;            // frameQueue(10), SpectrumDSP.cpp:47
constexpr size_t kStageSamples = 1 << 16;  // pinned staging slot, samples per channel
constexpr int kStageSlots = 4;
}

struct sgz_spectrum {
    Plan *plan = nullptr;
    std::mutex mu;                    // guards queue + (re)configuration
    hipStream_t stream = nullptr;
    // device audio history: [2C][cap], two buffers for compaction
    float *d_hist[2] = {nullptr, nullptr};
    int cur = 0;
    size_t cap = 0, fill = 0;
    uint32_t sinceLast = 0;           // processedSamplesSinceLastFrame
    float *d_mapped = nullptr, *d_state = nullptr, *d_lines = nullptr;
    uint8_t *d_cols = nullptr, *h_cols = nullptr;
    hipEvent_t colEvents[kQueueDepth] = {};
    std::deque<int> pending;          // slots with a column in flight / ready
    int nextSlot = 0;
    float *h_stage = nullptr;         // pinned [kStageSlots][2C][kStageSamples]
    hipEvent_t stageEvents[kStageSlots] = {};
    int stageSlot = 0;
    uint64_t dropped = 0;
};

static void freeHandle(sgz_spectrum *s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (float *p : {s->d_hist[0], s->d_hist[1], s->d_mapped, s->d_state, s->d_lines}) if (p) (void)hipFree(p);
    if (s->d_cols) (void)hipFree(s->d_cols);
    if (s->h_cols) (void)hipHostFree(s->h_cols);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    for (auto &e : s->colEvents) if (e) (void)hipEventDestroy(e);
    for (auto &e : s->stageEvents) if (e) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s->plan;
    delete s;
}

static sgz_status setup(sgz_spectrum *s, const sgz_spectrum_config *cfg)
{
    Plan *pl = new (std::nothrow) Plan();
    if (!pl) return fail(SGZ_ENOMEM, "out of memory");
    std::string err;
    sgz_status st = buildPlan(*cfg, *pl, err);
    if (st == SGZ_OK) st = uploadPlan(*pl, err);
    if (st != SGZ_OK) { delete pl; return fail(st, err); }
    if (!s->stream) SGZ_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    (void)hipStreamSynchronize(s->stream);
    delete s->plan;
    s->plan = pl;
    Plan &p = *pl;
    const size_t nch = size_t(2) * p.C;
    for (float **q : {&s->d_hist[0], &s->d_hist[1], &s->d_mapped, &s->d_state, &s->d_lines}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    if (s->d_cols) { (void)hipFree(s->d_cols); s->d_cols = nullptr; }
    if (s->h_cols) { (void)hipHostFree(s->h_cols); s->h_cols = nullptr; }
    s->cap = size_t(p.W) + 8 * std::max<size_t>(p.cfg.hop, kStageSamples);
    for (int b = 0; b < 2; ++b) SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_hist[b]), nch * s->cap * sizeof(float)));
    SGZ_HIP(hipMemsetAsync(s->d_hist[0], 0, nch * s->cap * sizeof(float), s->stream));
    s->cur = 0;
    s->fill = p.W;                    // history starts as W samples of silence (a full, zeroed ring)
    s->sinceLast = 0;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_mapped), size_t(p.C) * p.sides * p.P * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_state), stateN * sizeof(float)));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_lines), stateN * sizeof(float)));
    SGZ_HIP(hipMemsetAsync(s->d_state, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipMemsetAsync(s->d_lines, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_cols), size_t(kQueueDepth) * p.P * 4));
    SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_cols), size_t(kQueueDepth) * p.P * 4, hipHostMallocDefault));
    if (!s->h_stage) SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_stage), size_t(kStageSlots) * 32 * kStageSamples * sizeof(float), hipHostMallocDefault));
    for (auto &e : s->colEvents) if (!e) SGZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : s->stageEvents) if (!e) SGZ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->pending.clear();
    s->nextSlot = 0;
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
    std::lock_guard<std::mutex> lk(s->mu);
    return setup(s, cfg);
}

sgz_status sgz_spectrum_clear_state(sgz_spectrum *s)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    Plan &p = *s->plan;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    SGZ_HIP(hipMemsetAsync(s->d_state, 0, stateN * sizeof(float), s->stream));
    SGZ_HIP(hipMemsetAsync(s->d_lines, 0, stateN * sizeof(float), s->stream));
    return SGZ_OK;
}

// one frame over the W newest samples of the history
static sgz_status fireFrame(sgz_spectrum *s)
{
    Plan &p = *s->plan;
    const float *base = s->d_hist[s->cur] + (s->fill - p.W);
    sgz_status st = runStft(p, base, s->cap, 1, s->d_mapped, nullptr, nullptr, s->stream);
    if (st != SGZ_OK) return st;
    int slot = -1;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        if (int(s->pending.size()) < kQueueDepth) { slot = s->nextSlot; s->nextSlot = (s->nextSlot + 1) % kQueueDepth; }
        else s->dropped++;            // acquireFreeElement failed: frame dropped (SpectrumDSP.cpp:185-186)
    }
    uint8_t *d_col = slot >= 0 ? s->d_cols + size_t(slot) * p.P * 4 : nullptr;
    st = runDecayColour(p, s->d_mapped, 1, d_col, s->d_lines, s->d_state, s->stream);
    if (st != SGZ_OK) return st;
    if (slot >= 0) {
        SGZ_HIP(hipMemcpyAsync(s->h_cols + size_t(slot) * p.P * 4, d_col, size_t(p.P) * 4, hipMemcpyDeviceToHost, s->stream));
        SGZ_HIP(hipEventRecord(s->colEvents[slot], s->stream));
        std::lock_guard<std::mutex> lk(s->mu);
        s->pending.push_back(slot);
    }
    return SGZ_OK;
}

sgz_status sgz_spectrum_push(sgz_spectrum *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples)
{
    if (!s || !planar) return fail(SGZ_EINVAL, "null argument");
    Plan &p = *s->plan;
    if (num_channels != 2 * p.C) return fail(SGZ_EINVAL, "num_channels must equal 2*num_pairs (SpectrumDSP.cpp:65-72)");
    uint32_t done = 0;
    while (done < nsamples) {
        // consume up to the next frame boundary (TransformDSP.inl:1172-1183)
        const uint32_t remaining = s->sinceLast >= p.cfg.hop ? 0 : p.cfg.hop - s->sinceLast;
        uint32_t m = std::min<uint32_t>(nsamples - done, remaining ? remaining : 1);
        m = std::min<uint32_t>(m, uint32_t(kStageSamples));
        if (s->fill + m > s->cap) {          // compact: keep the W newest samples
            const int nxt = s->cur ^ 1;
            SGZ_HIP(hipMemcpy2DAsync(s->d_hist[nxt], s->cap * sizeof(float), s->d_hist[s->cur] + (s->fill - p.W),
                                     s->cap * sizeof(float), size_t(p.W) * sizeof(float), num_channels,
                                     hipMemcpyDeviceToDevice, s->stream));
            s->cur = nxt;
            s->fill = p.W;
        }
        // stage through pinned memory so the copy is truly asynchronous
        const int slot = s->stageSlot;
        s->stageSlot = (s->stageSlot + 1) % kStageSlots;
        (void)hipEventSynchronize(s->stageEvents[slot]);     // slot reuse: normally long complete
        float *stage = s->h_stage + size_t(slot) * 32 * kStageSamples;
        for (uint32_t c = 0; c < num_channels; ++c) std::memcpy(stage + size_t(c) * m, planar[c] + done, size_t(m) * sizeof(float));
        SGZ_HIP(hipMemcpy2DAsync(s->d_hist[s->cur] + s->fill, s->cap * sizeof(float), stage, size_t(m) * sizeof(float),
                                 size_t(m) * sizeof(float), num_channels, hipMemcpyHostToDevice, s->stream));
        SGZ_HIP(hipEventRecord(s->stageEvents[slot], s->stream));
        s->fill += m;
        s->sinceLast += m;
        done += m;
        if (s->sinceLast >= p.cfg.hop) {     // :1185
            sgz_status st = fireFrame(s);
            if (st != SGZ_OK) return st;
            s->sinceLast = 0;
        }
    }
    return SGZ_OK;
}

sgz_status sgz_spectrum_pop_column(sgz_spectrum *s, uint8_t *rgba, uint32_t *axis_points)
{
    if (!s || !rgba) return fail(SGZ_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->pending.empty()) return SGZ_EMPTY;
    const int slot = s->pending.front();
    const hipError_t q = hipEventQuery(s->colEvents[slot]);
    if (q == hipErrorNotReady) return SGZ_EMPTY;
    if (q != hipSuccess) return hipFail(q, "hipEventQuery");
    const Plan &p = *s->plan;
    std::memcpy(rgba, s->h_cols + size_t(slot) * p.P * 4, size_t(p.P) * 4);
    if (axis_points) *axis_points = p.P;
    s->pending.pop_front();
    return SGZ_OK;
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

}  // extern "C"
