// api.hip -- the extern "C" shim of libsgz.so (include/sgz.h): plan lifetime, the offline/batch spectrogram
// entry points, the per-stage hooks the parity tests call, and the real-time push/pop handle.
// There is NO CPU fallback: without a gfx950 device every compute entry point fails with SGZ_EHIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "runtime.hpp"
#include "trace.hpp"

namespace sgz {

thread_local std::string g_lastError;
#ifdef SGZ_DEBUG
uint32_t g_ablate = 0;      // tools/ablate.py
#endif

sgz_status fail(sgz_status st, const std::string &msg)
{
    g_lastError = msg;
    return st;
}
sgz_status hipFail(hipError_t e, const char *what)
{
    g_lastError = std::string(what) + ": " + hipGetErrorString(e);
    return SGZ_EHIP;
}

Plan::~Plan()
{
    // best effort; ignore errors on teardown
    void *ptrs[] = {d_window, d_slope, d_colourTables, d_weights, d_weights11, d_recsReal, d_realLowPixels, d_low, d_tw1, d_tw2, d_twN, d_tw1odd, d_recs, d_items, d_mapped, d_agg, d_scratch,
                    d_stateCopy, d_work0, d_work1, d_binsWork, d_halfBins, d_dcPixels, d_dcWork, d_phaseType, d_phaseNorm, d_phaseWork, d_shard, d_twReal1, d_twRealPost, d_tw2Full, d_tw16, d_twPost16, d_windowHalf, d_winPhase, d_winPhaseT, d_ny, d_nyBest, d_chunkEnds, d_chunkReBase, d_chunkRec, d_weights12, d_resCoeff, d_resPow, d_resPowB, d_resPowBLo, d_resW1, d_resW2, d_resW1b, d_resTile, d_resGain, d_resState, d_resLocal};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (void *p : {(void *)d_hostAudio, (void *)d_hostRgba, (void *)d_hostLines})
        if (p) (void)hipFree(p);
    for (void *e : hostEv) if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    for (void *e : shardEv) if (e) (void)hipEventDestroy(static_cast<hipEvent_t>(e));
    if (shardStream) (void)hipStreamDestroy(static_cast<hipStream_t>(shardStream));
    if (hostStream) (void)hipStreamDestroy(static_cast<hipStream_t>(hostStream));
}

template <typename T>
static sgz_status uploadVec(const std::vector<T> &v, T **dst)
{
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    if (v.empty()) return SGZ_OK;
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(T)));
    SGZ_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return SGZ_OK;
}

sgz_status uploadPlan(Plan &p, std::string &err)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        err = "no HIP device visible (libsgz has no CPU fallback)";
        return SGZ_EHIP;
    }
    sgz_status st;
    if ((st = uploadVec(p.window, &p.d_window)) != SGZ_OK) return st;
    if ((st = uploadVec(p.slope, &p.d_slope)) != SGZ_OK) return st;
    if ((st = uploadVec(p.colourTables, &p.d_colourTables)) != SGZ_OK) return st;
    if ((st = uploadVec(p.weights, &p.d_weights)) != SGZ_OK) return st;
    if ((st = uploadVec(p.weights11, &p.d_weights11)) != SGZ_OK) return st;
    if ((st = uploadVec(p.recsReal, &p.d_recsReal)) != SGZ_OK) return st;
    if ((st = uploadVec(p.realLowPixels, &p.d_realLowPixels)) != SGZ_OK) return st;
    if ((st = uploadVec(p.tw1, &p.d_tw1)) != SGZ_OK) return st;
    if ((st = uploadVec(p.tw2, &p.d_tw2)) != SGZ_OK) return st;
    if ((st = uploadVec(p.twN, &p.d_twN)) != SGZ_OK) return st;
    if ((st = uploadVec(p.tw1odd, &p.d_tw1odd)) != SGZ_OK) return st;
    if ((st = uploadVec(p.twReal1, &p.d_twReal1)) != SGZ_OK) return st;
    if ((st = uploadVec(p.twRealPost, &p.d_twRealPost)) != SGZ_OK) return st;
    if ((st = uploadVec(p.tw2Full, &p.d_tw2Full)) != SGZ_OK) return st;
    if ((st = uploadVec(p.tw16, &p.d_tw16)) != SGZ_OK) return st;
    if ((st = uploadVec(p.windowHalf, &p.d_windowHalf)) != SGZ_OK) return st;
    if ((st = uploadVec(p.twPost16, &p.d_twPost16)) != SGZ_OK) return st;
    if ((st = uploadVec(p.winPhase, &p.d_winPhase)) != SGZ_OK) return st;
    if ((st = uploadVec(p.winPhaseT, &p.d_winPhaseT)) != SGZ_OK) return st;
    if ((st = uploadVec(p.chunkEnds, &p.d_chunkEnds)) != SGZ_OK) return st;
    if ((st = uploadVec(p.chunkReBase, &p.d_chunkReBase)) != SGZ_OK) return st;
    if ((st = uploadVec(p.chunkRec, &p.d_chunkRec)) != SGZ_OK) return st;
    if ((st = uploadVec(p.weights12, &p.d_weights12)) != SGZ_OK) return st;
    if ((st = uploadVec(p.dcPixels, &p.d_dcPixels)) != SGZ_OK) return st;
    if ((st = uploadVec(p.recs, &p.d_recs)) != SGZ_OK) return st;
    if ((st = uploadVec(p.items, &p.d_items)) != SGZ_OK) return st;
    if ((st = uploadVec(p.phaseType, &p.d_phaseType)) != SGZ_OK) return st;
    if ((st = uploadVec(p.phaseNorm, &p.d_phaseNorm)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resCoeff, &p.d_resCoeff)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resPow, &p.d_resPow)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resPowB, &p.d_resPowB)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resPowBLo, &p.d_resPowBLo)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resW1, &p.d_resW1)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resW2, &p.d_resW2)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resW1b, &p.d_resW1b)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resTile, &p.d_resTile)) != SGZ_OK) return st;
    if ((st = uploadVec(p.resGain, &p.d_resGain)) != SGZ_OK) return st;
    if (isResonator(p)) {                                  // the resonators start from rest (TransformPair.h:183 resetState)
        if (p.d_resState) { (void)hipFree(p.d_resState); p.d_resState = nullptr; }
        const size_t n = size_t(p.C) * 2 * size_t(p.resV) * p.P * 2 * sizeof(float);
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_resState), n));
        SGZ_HIP(hipMemset(p.d_resState, 0, n));
    }
    (void)hipGetDevice(&p.device);
    p.uploaded = true;
    return SGZ_OK;
}

sgz_status ensureCap(float **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return SGZ_OK;
    if (*buf) { (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(buf), need * sizeof(float)));
    *cap = need;
    return SGZ_OK;
}

int numCUs()
{
    // cached per device: sgz_set_device may move a host thread between devices of different sizes
    static std::atomic<int> cache[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool slot = dev >= 0 && dev < 64;
    int cus = slot ? cache[dev].load(std::memory_order_relaxed) : 0;
    if (!cus) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (slot) cache[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}

// K_A launch over `frames` frames of `planar`; writes mapped/bins as requested.
static StftParams fillStftParams(Plan &p, const float *d_planar, size_t chStride, long frames, float *d_mapped, float *d_binsOut,
                                 const float *d_binsIn, unsigned long long *d_phaseClock)
{
    StftParams prm{};
    prm.planar = d_planar;
    prm.chStride = chStride;
    prm.frames = frames;
    prm.hop = p.cfg.hop; prm.W = p.W; prm.P = p.P; prm.C = p.C;
    prm.sides = uint32_t(p.sides); prm.mode = p.cfg.channel_mode;
    prm.window = p.d_window;
    prm.winPhase = p.optFetchWindow ? nullptr : reinterpret_cast<const float2 *>(p.d_winPhaseT); prm.winP0 = p.winP0; prm.winP1 = p.winP1;
    prm.tw1 = reinterpret_cast<const float2 *>(p.d_tw1);
    prm.tw2 = reinterpret_cast<const float2 *>(p.d_tw2);
    prm.tw1odd = reinterpret_cast<const float2 *>(p.d_tw1odd);
    prm.dcPixels = p.d_dcPixels; prm.nDcPixels = uint32_t(p.dcPixels.size());
    prm.recs = p.d_recs; prm.weights = p.d_weights; prm.weights11 = p.d_weights11;
    prm.items = p.d_items; prm.nItems = uint32_t(p.items.size()); prm.nItemsLeft = p.nItemsLeft;
    prm.invSize = p.scalars.invSize;
    prm.roundSize = uint32_t(numCUs());
    prm.mapped = d_mapped; prm.binsOut = d_binsOut; prm.binsIn = d_binsIn; prm.phaseClock = d_phaseClock;
#ifdef SGZ_DEBUG
    prm.ablate = g_ablate;
#endif
    return prm;
}

// Complex mode, generic / halves path: csf[0] of every task of a slab
static sgz_status ensureDcWork(Plan &p, size_t slab)
{
    if (p.dcSlab >= slab) return SGZ_OK;
    if (p.d_dcWork) { (void)hipFree(p.d_dcWork); p.d_dcWork = nullptr; }
    SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_dcWork), slab * 16 * 2 * sizeof(float)));          // [task][kSpecBins] float2
    p.dcSlab = slab;
    return SGZ_OK;
}

// the channel-split launch's late-pixel fields alone (realLateKernel as a launch of its own, runDecayColour's fallback)
static RealParams fillRealLate(Plan &p, long frames, float *d_mapped)
{
    RealParams rp{};
    rp.frames = frames; rp.C = p.C; rp.P = p.P; rp.mode = p.cfg.channel_mode;
    rp.recsFull = p.d_recs; rp.weights = p.d_weights;
    rp.low = p.d_low; rp.lowPixels = p.d_realLowPixels; rp.lowCount[0] = p.realLowCount[0]; rp.lowCount[1] = p.realLowCount[1];
    rp.invSize = p.scalars.invSize; rp.mapped = d_mapped;
    rp.ny = p.d_ny; rp.nyBest = p.d_nyBest; rp.fixFrom[0] = p.realFixFrom[0]; rp.fixFrom[1] = p.realFixFrom[1];
    return rp;
}

// RSNT: the resonators advance over frames x hop samples starting at d_planar, continuing from the state the plan carries
sgz_status resetResonator(Plan &p, hipStream_t stream)
{
    if (!isResonator(p) || !p.d_resState) return SGZ_OK;
    SGZ_HIP(hipMemsetAsync(p.d_resState, 0, size_t(p.C) * 2 * size_t(p.resV) * p.P * 2 * sizeof(float), stream));
    return SGZ_OK;
}

// hopOverride (line-graph mode of the real-time handle): ONE "frame" of that many samples -- the resonators advance over a whole host
// block (TransformDSP.inl:1206-1209), sample by sample from the carried state, and d_mapped receives the windowed state afterwards
static sgz_status fillResParams(Plan &p, const float *d_planar, size_t chStride, long frames, float *d_mapped, uint32_t hopOverride, ResParams &r)
{
    r = ResParams{};
    r.planar = d_planar; r.chStride = chStride; r.frames = frames;
    r.hop = hopOverride ? hopOverride : p.cfg.hop; r.C = p.C; r.P = p.P; r.mode = p.cfg.channel_mode;
    r.V = p.resV; r.signals = p.stateChannels; r.sides = p.sides; r.firstContinues = true;
    r.coeff = reinterpret_cast<const float2 *>(p.d_resCoeff);
    r.cpow = reinterpret_cast<const float4 *>(p.d_resPow);
    r.cpowB = reinterpret_cast<const float2 *>(p.d_resPowB);
    r.cpowBLo = reinterpret_cast<const float2 *>(p.d_resPowBLo);
    r.matrixForm = p.optMatrixResonator;
    r.w1 = p.optMatrixResonator ? reinterpret_cast<const float2 *>(p.d_resW1) : nullptr;       // (null: hop is not a multiple of 1024)
    r.w2 = reinterpret_cast<const float2 *>(p.d_resW2);
    r.w1b = reinterpret_cast<const uint4 *>(p.d_resW1b);
    r.tilePow = reinterpret_cast<const float4 *>(p.d_resTile);
    r.gain = p.d_resGain;
    for (int v = 0; v < 9; ++v) r.weights[v] = p.resWeights[v];
    r.state = reinterpret_cast<float2 *>(p.d_resState);
    const size_t perFrame = size_t(p.C) * size_t(r.signals) * size_t(r.V) * p.P;                 // complex values; the segments' end states follow the frames
    sgz_status st = ensureCap(&p.d_resLocal, &p.resLocalCap, (size_t(frames) + size_t(kResSegments)) * perFrame * 2);
    if (st != SGZ_OK) return st;
    r.local = reinterpret_cast<float2 *>(p.d_resLocal);
    r.segEnd = r.local + size_t(frames) * perFrame;
    r.mapped = d_mapped;
    return SGZ_OK;
}

static sgz_status runResonator(Plan &p, const float *d_planar, size_t chStride, long frames, float *d_mapped, hipStream_t stream, uint32_t hopOverride = 0,
                               bool skipWindow = false)
{
    if (frames <= 0) return SGZ_OK;
    if (!d_mapped) return fail(SGZ_EUNSUPPORTED, "the resonator algorithm has no transform bins: ask for mapped values");
    if (hopOverride && frames != 1) return fail(SGZ_EINVAL, "a block advance of the resonators is one frame");
    // long renders go in slabs of frames: the per-frame states between the kernels ([frames][C][signals][V][P] complex) are V times the
    // mapped buffer -- a ten-minute flat-top render at hop 1024 would ask for gigabytes.  A slab continues the state the slab before it
    // left in the plan, exactly as a render continues the carried state.
    if (!skipWindow && !hopOverride) {
        const size_t perFrame = size_t(p.C) * size_t(p.stateChannels) * size_t(p.resV) * p.P * 2 * sizeof(float);
        const long slab = p.optResonatorSlab ? long(p.optResonatorSlab) : long(std::max<size_t>(64, (size_t(256) << 20) / perFrame));
        if (frames > slab) {
            for (long f0 = 0; f0 < frames; f0 += slab) {
                const long nf = std::min(slab, frames - f0);
                const sgz_status st = runResonator(p, d_planar + size_t(f0) * p.cfg.hop, chStride, nf, d_mapped + size_t(f0) * p.C * p.sides * p.P, stream);
                if (st != SGZ_OK) return st;
            }
            return SGZ_OK;
        }
    }
    ResParams r;
    if (sgz_status st = fillResParams(p, d_planar, chStride, frames, d_mapped, hopOverride, r); st != SGZ_OK) return st;
    r.skipWindow = skipWindow;
    SGZ_HIP(launchResonator(r, stream));
    return SGZ_OK;
}

// The two halves of a sharded RSNT render (sharded.hip): the resonators over this rank's chunk FROM REST, chained, windows not yet
// applied (the plan's state = the end state from rest, what the ranks exchange); then the entering state folded from the gathered
// end states is added to every frame and the window kernel fills d_mapped.
// The per-frame states of a rank's WHOLE chunk stay on the device between the two halves of a sharded RSNT render (runResonatorJoin adds
// the carry to every frame of them), so that path cannot go in slabs like a plain render.  Its bound is an option of its own,
// SGZ_OPT_RESONATOR_SHARD_BOUND (default: 8 GiB worth of frames -- a rank's chunk of a sharded job is sized for the device, not for a
// plugin's working set); SGZ_OPT_RESONATOR_SLAB, the plain render's working-set knob, plays no part here.  A chunk above the bound is
// refused, on every rank alike, before anything is allocated or exchanged.
sgz_status checkResonatorShardBound(const Plan &p, long frames)
{
    const size_t perFrame = size_t(p.C) * size_t(p.stateChannels) * size_t(p.resV) * p.P * 2 * sizeof(float);
    const size_t bound = p.optResonatorShardBound ? size_t(p.optResonatorShardBound) : std::max<size_t>(64, (size_t(8) << 30) / std::max<size_t>(perFrame, 1));
    if (frames > 0 && size_t(frames) > bound)
        return fail(SGZ_EUNSUPPORTED, "sharded RSNT render: a rank's chunk of " + std::to_string(frames) + " frames holds " +
                                          std::to_string((size_t(frames) * perFrame) >> 20) + " MiB of per-frame resonator states between the two halves of the render, above the bound of " +
                                          std::to_string(bound) + " frames (SGZ_OPT_RESONATOR_SHARD_BOUND; default 8 GiB): use more ranks or a shorter buffer per call");
    return SGZ_OK;
}
sgz_status runResonatorFromRest(Plan &p, const float *d_planar, size_t chStride, long frames, float *d_mapped, hipStream_t stream)
{
    if (sgz_status st = checkResonatorShardBound(p, frames); st != SGZ_OK) return st;
    if (sgz_status st = resetResonator(p, stream); st != SGZ_OK) return st;
    return runResonator(p, d_planar, chStride, frames, d_mapped, stream, 0, /*skipWindow=*/true);
}
sgz_status runResonatorJoin(Plan &p, long frames, float *d_mapped, const float *d_allEnd, const long long *framesPerRank, uint32_t world, uint32_t rank,
                            float *d_carry, hipStream_t stream)
{
    if (frames <= 0) return SGZ_OK;
    ResParams r;
    if (sgz_status st = fillResParams(p, nullptr, 0, frames, d_mapped, 0, r); st != SGZ_OK) return st;
    const float2 *carry = nullptr;
    if (rank > 0) {
        SGZ_HIP(launchResonatorFold(r, reinterpret_cast<const float2 *>(d_allEnd), framesPerRank, world, rank, reinterpret_cast<float2 *>(d_carry), stream));
        carry = reinterpret_cast<const float2 *>(d_carry);
    }
    SGZ_HIP(launchResonatorCarry(r, carry, stream));
    return SGZ_OK;
}

sgz_status runResonatorAdvance(Plan &p, const float *d_planar, size_t chStride, uint32_t nsamples, float *d_mapped, hipStream_t stream)
{
    if (!isResonator(p)) return fail(SGZ_EINVAL, "not a resonator plan");
    if (nsamples == 0) return SGZ_OK;
    return runResonator(p, d_planar, chStride, 1, d_mapped, stream, nsamples);
}

sgz_status ensureSecondStream(Plan &p)
{
    if (!p.shardStream) { hipStream_t cs; SGZ_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking)); p.shardStream = cs; }
    for (void *&e : p.shardEv)
        if (!e) { hipEvent_t ne; SGZ_HIP(hipEventCreateWithFlags(&ne, hipEventDisableTiming)); e = ne; }
    return SGZ_OK;
}

sgz_status runStft(Plan &p, const float *d_planar, size_t chStride, long frames, float *d_mapped,
                          float *d_binsOut, const float *d_binsIn, hipStream_t stream, unsigned long long *d_phaseClock, bool deferLate)
{
    TraceRange range("sgz::runStft (K_A)");
    p.lateDeferred = nullptr; p.lateFrames = 0;
    if (isResonator(p)) {
        if (d_binsOut || d_binsIn) return fail(SGZ_EUNSUPPORTED, "the resonator algorithm has no transform bins");
        return runResonator(p, d_planar, chStride, frames, d_mapped, stream);
    }
    const bool phase = p.cfg.channel_mode == SGZ_CH_PHASE;
    StftParams prm = fillStftParams(p, d_planar, chStride, frames, d_mapped, d_binsOut, d_binsIn, d_phaseClock);
    const long tasks = frames * long(p.C);
    if (tasks <= 0) return SGZ_OK;
    if (tasks > 0x7fffffffL) return fail(SGZ_EINVAL, "too many (frame, pair) tasks for one launch");
    // Every eligible plan runs the channel-split form (spectrum_real.hip).  Since the chunk-scan map and the late-pixel pass replaced
    // the piece list and the in-kernel pair exchange it wins at every launch size on MI355X (N = 32768: 348 frames 36 us against 44 us
    // for the whole-frame kernel, 2784 frame-pairs 212 us against 238 us); sgz_plan_set_option(SGZ_OPT_CHANNEL_SPLIT, 0) keeps a plan off it.
    // (rows need no more than their natural 4-byte alignment: gfx950's global_load_dwordx2 takes dword-aligned addresses, measured
    // bit-identical and within 2 % of 8-byte aligned rows, tools/unaligned_probe.py -- so the choice of kernel never depends on the layout)
    if (p.optChannelSplit && ((p.realMono && d_binsIn == nullptr) || p.realSplit)) {
        // Separate mode, N = 32768 / 65536, full window: one workgroup per (frame, pair, channel) (spectrum_real.hip)
        const size_t units = size_t(tasks) * 2;
        if (p.nyCap < units) {
            for (float **b : {&p.d_ny, &p.d_nyBest, &p.d_low}) if (*b) { (void)hipFree(*b); *b = nullptr; }
            p.nyCap = 0;
            // what the channel workgroups leave for realLateKernel (spectrum_real.hip): Nyquist bins, winning squares of the top pixels, lowest bins
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_nyBest), units * 64 * sizeof(float)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_ny), units * sizeof(float)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_low), units * kLowBins * sizeof(float)));
            p.nyCap = units;
        }
        RealParams rp{};
        rp.planar = d_planar; rp.chStride = chStride; rp.frames = frames;
        rp.hop = p.cfg.hop; rp.C = p.C; rp.P = p.P; rp.mode = p.cfg.channel_mode;
        rp.window = p.d_windowHalf;                                       // these kernels transform x w / 2 (real_common.hpp realBinMag)
        rp.winPhase = p.optFetchWindow ? nullptr : reinterpret_cast<const float4 *>(p.d_winPhase); rp.winP0 = 0.5f * p.winP0; rp.winP1 = 0.5f * p.winP1;
        rp.tw1 = reinterpret_cast<const float2 *>(p.d_twReal1);
        rp.tw2 = reinterpret_cast<const float2 *>(p.d_tw2);
        rp.twPost = reinterpret_cast<const float2 *>(p.d_twRealPost);
        rp.tw2Full = reinterpret_cast<const float4 *>(p.d_tw2Full);
        rp.tw16 = p.optWideGroups ? reinterpret_cast<const float4 *>(p.d_tw16) : nullptr; rp.twPost16 = reinterpret_cast<const float2 *>(p.d_twPost16);
        rp.recs = p.d_recsReal ? p.d_recsReal : p.d_recs; rp.recsFull = p.d_recs; rp.weights = p.d_weights;
        rp.chunkEnds = p.d_chunkEnds; rp.chunkReBase = p.d_chunkReBase; rp.chunkRec = p.d_chunkRec; rp.weights12 = p.d_weights12;
        rp.chunkSlots[0] = p.chunkSlots[0]; rp.chunkSlots[1] = p.chunkSlots[1];
        rp.low = p.d_low; rp.lowPixels = p.d_realLowPixels; rp.lowCount[0] = p.realLowCount[0]; rp.lowCount[1] = p.realLowCount[1];
        rp.invSize = p.scalars.invSize;
        rp.mapped = d_mapped; rp.binsOut = d_binsOut; rp.binsIn = d_binsIn;
        rp.ny = p.d_ny; rp.nyBest = p.d_nyBest;
        rp.fixFrom[0] = p.realFixFrom[0]; rp.fixFrom[1] = p.realFixFrom[1];
        // the pixels that need both channels: realLateKernel behind the channel workgroups, or -- an image-only render, no low pixels --
        // K_B's fused kernel while it loads the magnitudes (then the step stays at two launches)
        if (deferLate && !p.realMono && d_mapped && !d_binsOut && p.realLowCount[0] + p.realLowCount[1] == 0) { rp.lateInNext = 1u; p.lateDeferred = d_mapped; p.lateFrames = frames; }
        rp.roundSize = uint32_t(numCUs()) * (p.N == 16384 ? 4u : p.N == 32768 ? 2u : 1u);   // workgroups a CU holds at once
        rp.pipelined = p.optPipelined ? 1u : 0u;
#ifdef SGZ_DEBUG
        rp.phaseClock = d_phaseClock; rp.clkUnit = g_ablate >> 16;
#endif
        SGZ_HIP(launchStftReal(rp, p.N, stream));
        return SGZ_OK;
    }
    if (p.halves && d_binsIn == nullptr) {
        // N = 2 R^3: half-frame workgroups -> csf magnitudes in HBM (a slab of tasks at a time) -> mapSideKernel / genericMap
        const size_t perTask = size_t(p.N) + 1;
        long slab = long(std::max<size_t>(1, (size_t(32) << 20) / perTask));       // 128 MiB of bins: stays close to the last-level cache
        slab = std::max<long>(1, slab / long(p.C)) * long(p.C);                  // whole frames: the kernel walks a slab pair-major
        slab = std::min<long>(slab, tasks);
        if (d_binsOut == nullptr && p.binsSlab < size_t(slab)) {
            if (p.d_halfBins) { (void)hipFree(p.d_halfBins); p.d_halfBins = nullptr; }
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_halfBins), size_t(slab) * perTask * sizeof(float)));
            p.binsSlab = size_t(slab);
        }
        if (prm.nDcPixels) { sgz_status st = ensureDcWork(p, size_t(slab)); if (st != SGZ_OK) return st; prm.dcOut = reinterpret_cast<float2 *>(p.d_dcWork); }
        for (long t0 = 0; t0 < tasks; t0 += slab) {
            const long nt = std::min(slab, tasks - t0);
            float *bins = d_binsOut ? d_binsOut + size_t(t0) * perTask : p.d_halfBins;
            prm.taskBase = t0;
            prm.frames = nt / long(p.C);
            prm.binsOut = bins;
            // mapSideKernel reads the two halves' bins as they are produced (two contiguous arrays); the test hook and the
            // generic map kernel (tall views, Complex) want csf order
            prm.binsSplit = (d_binsOut == nullptr && p.sideMapOk && mapSidesFit(prm, p.N)) ? 1u : 0u;
            SGZ_HIP(launchStftHalves(prm, p.N, int(2 * nt), stream));
            if (d_mapped) {
                float *out = d_mapped + size_t(t0) * p.sides * p.P;
                if (prm.binsSplit) SGZ_HIP(launchMapSides(prm, p.N, bins, nt, out, stream));
                else SGZ_HIP(launchGenericMap(prm, p.N, bins, nt, out, stream));
                SGZ_HIP(launchComplexDcFix(prm, p.N, bins, prm.dcOut, nt, out, stream));
            }
        }
        return SGZ_OK;
    }
    if (!p.fused) {
        // generic multi-kernel path (any power-of-two N): work buffers for a slab of tasks, <= 256 MiB each
        const size_t perTask = size_t(p.N) * 2;                                  // floats of one complex buffer
        long slab = long(std::max<size_t>(1, (size_t(64) << 20) / perTask));      // 64 Mi floats = 256 MiB
        slab = std::max<long>(1, slab / long(p.C)) * long(p.C);                  // whole frames (the in-register FFT walks a slab pair-major)
        slab = std::min<long>(slab, tasks);
        if (p.workSlab < size_t(slab)) {
            for (float **b : {&p.d_work0, &p.d_work1, &p.d_binsWork}) if (*b) { (void)hipFree(*b); *b = nullptr; }
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_work0), size_t(slab) * perTask * sizeof(float)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_work1), size_t(slab) * perTask * sizeof(float)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&p.d_binsWork), size_t(slab) * (size_t(p.N) + 1) * sizeof(float) * (phase ? 2 : 1)));
            p.workSlab = size_t(slab);
        }
        if (prm.nDcPixels) { sgz_status st = ensureDcWork(p, p.workSlab); if (st != SGZ_OK) return st; prm.dcOut = reinterpret_cast<float2 *>(p.d_dcWork); }
        PhaseTables ph{};
        if (phase) {
            // Phase keeps the bins complex; the float test hooks carry float2 data in this mode (sgz.h, stage hooks)
            ph.type = p.d_phaseType; ph.norm = p.d_phaseNorm; ph.normFinal = p.phaseNormFinal;
            ph.filtered = p.cfg.bin_interp != SGZ_INTERP_NONE;
            ph.fusedFft = p.phaseFusedFft ? 1u : 0u;
            ph.csfOut = reinterpret_cast<float2 *>(d_binsOut);
            ph.csfIn = reinterpret_cast<const float2 *>(d_binsIn);
        }
        SGZ_HIP(launchGeneric(prm, p.N, reinterpret_cast<const float2 *>(p.d_twN), reinterpret_cast<float2 *>(p.d_work0),
                              reinterpret_cast<float2 *>(p.d_work1), p.d_binsWork, long(p.workSlab), stream, phase ? &ph : nullptr,
                              p.sideMapOk && !phase));
        return SGZ_OK;
    }
    const int grid = int(tasks);                 // one workgroup per (frame, pair); the dispatcher refills CUs
    SGZ_HIP(launchStftMap(prm, p.N, grid, stream));
    return SGZ_OK;
}

// DecayParams of one K_B pass (snapshots the carry-in state when the pass both reads and writes it)
static sgz_status fillDecayParams(Plan &p, const float *d_mapped, long frames, uint8_t *d_rgba, float *d_lines, float *d_state,
                                  hipStream_t stream, DecayParams &prm)
{
    prm = DecayParams{};
    prm.mapped = d_mapped;
    prm.frames = frames;
    prm.P = p.P; prm.C = p.C; prm.sides = uint32_t(p.sides);
    prm.chunk = 8;
    prm.numChunks = uint32_t((frames + prm.chunk - 1) / prm.chunk);
    prm.slope = p.d_slope;
    prm.colourTables = p.d_colourTables;
    prm.sc = p.scalars;
    prm.state = d_state; prm.stateIn = d_state; prm.rgba = d_rgba; prm.lines = d_lines;
    prm.colourOnly = (!d_state && !d_lines && d_rgba) ? 1u : 0u;
    prm.magScale = p.cfg.channel_mode == SGZ_CH_PHASE ? 0.5f : 1.0f;
    if (d_state && frames > 1) {
        // frame 0's threads read the carry-in while the last frame's threads write the new state: snapshot it
        const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
        sgz_status st0 = ensureCap(&p.d_stateCopy, &p.stateCopyCap, stateN);
        if (st0 != SGZ_OK) return st0;
        if (prm.numChunks > 1 && p.cfg.channel_mode != SGZ_CH_PHASE) {
            // two launches: the scan launch reads the live state and stashes what it read, the emit launch reads the stash (no copy launch)
            prm.stateStash = p.d_stateCopy;
        } else {
            SGZ_HIP(hipMemcpyAsync(p.d_stateCopy, d_state, stateN * sizeof(float), hipMemcpyDeviceToDevice, stream));
            prm.stateIn = p.d_stateCopy;
        }
    }
    return SGZ_OK;
}

sgz_status runDecayColour(Plan &p, const float *d_mapped, long frames, uint8_t *d_rgba, float *d_lines,
                                 float *d_state, hipStream_t stream, bool magnitudeOnly)
{
    // the channel-split K_A in front of this call may have left its late pixels to whoever reads its magnitudes next (runStft, deferLate):
    // taken over -- buffer and frame count -- before anything can return, so that no later call completes them on another launch's tables
    TraceRange range("sgz::runDecayColour (K_B)");
    float *pending = const_cast<float *>(p.lateDeferred);
    const long pendingFrames = p.lateFrames;
    p.lateDeferred = nullptr; p.lateFrames = 0;
    if (frames <= 0) return SGZ_OK;
    p.aggMapped = nullptr;                                   // whatever d_agg held is about to be overwritten (or left stale by a fused launch)
    DecayParams prm;
    sgz_status stp = fillDecayParams(p, d_mapped, frames, d_rgba, d_lines, d_state, stream, prm);
    if (stp != SGZ_OK) return stp;
    const bool noFused = !p.optFusedColour;                                  // (sgz_plan_set_option: A/B switch for measurements)
    prm.fusedPixels = p.optFusedPixels;
    // the one-launch form of the step with line results / state (one pair, 2 .. 64 chunks; not the scan half of the two-step K_B,
    // whose aggregates the emit half needs in HBM)
    const bool fullFused = !noFused && !magnitudeOnly && p.cfg.channel_mode != SGZ_CH_PHASE && prm.numChunks > 1 && decayFullFusedApplies(prm);
    if (pending) {
        if (d_mapped == pending && frames == pendingFrames && !noFused && p.cfg.channel_mode != SGZ_CH_PHASE && (decayColourFusedApplies(prm) || fullFused)) {
            // the fused colour kernel completes those pixels as it loads the magnitudes (spectrum_post.hip): the step stays at two
            // launches.  (Tried for the scan / emit kernels as well: the extra loads sit on their critical path, +3 us each,
            // against 2-5 us for realLateKernel as a launch of its own.)
            prm.late = LateFix{p.d_ny, p.d_nyBest, p.realFixFrom[0], p.realFixFrom[1], p.P, p.scalars.invSize, 0u};
            prm.hasLate = 1u;
        } else {
            SGZ_HIP(launchRealLate(fillRealLate(p, pendingFrames, pending), p.N, stream));
        }
    }
    if (p.cfg.channel_mode == SGZ_CH_PHASE && !magnitudeOnly) {
        // colour column only: the image reads the main graph's magnitude state alone (decayPhaseColourKernel), a plain peak decay of
        // plane 0 x 0.5 -- the chunked exact scan of the fused kernel instead of the sequential walk the phase smoother needs
        if (!noFused && decayColourFusedApplies(prm)) {
            SGZ_HIP(launchDecayColourFused(prm, stream));
            return SGZ_OK;
        }
        sgz_status stw = ensureCap(&p.d_phaseWork, &p.phaseWorkCap, size_t(frames) * p.C * p.P);
        if (stw != SGZ_OK) return stw;
        SGZ_HIP(launchDecayPhase(prm, p.d_phaseWork, stream));
        return SGZ_OK;
    }
    if (!noFused && decayColourFusedApplies(prm)) {
        SGZ_HIP(launchDecayColourFused(prm, stream));
        return SGZ_OK;
    }
    if (fullFused) {
        prm.stateStash = nullptr;                              // (the kernel parks its pixels' carry-in in LDS before anything writes the state)
        SGZ_HIP(launchDecayFullFused(prm, stream));
        return SGZ_OK;
    }
    if (prm.numChunks > 1) {
        const size_t need = size_t(prm.numChunks) * p.C * p.sides * SGZ_NUM_GRAPHS * p.P;
        sgz_status st = ensureCap(&p.d_agg, &p.aggCap, need);
        if (st != SGZ_OK) return st;
        prm.agg = p.d_agg;
        SGZ_HIP(launchDecayLocalCarry(prm, stream));
        p.aggMapped = d_mapped; p.aggFrames = frames;        // d_agg now holds the chunk aggregates of exactly this scan
        if (prm.stateStash) { prm.stateIn = prm.stateStash; prm.stateStash = nullptr; }
    }
    SGZ_HIP(launchDecayEmit(prm, stream));
    return SGZ_OK;
}

// second half of the two-step K_B (sgz_stage_decay_scan / sgz_stage_decay_emit): the aggregates of the preceding zero-carry scan of
// the same (mapped, frames) are still in p.d_agg; fold the true carry-in into them and emit
sgz_status runDecayEmitWithCarry(Plan &p, const float *d_mapped, long frames, const float *d_carry, uint8_t *d_rgba, float *d_lines,
                                 float *d_stateOut, hipStream_t stream)
{
    if (frames <= 0) return SGZ_OK;
    DecayParams prm{};
    prm.mapped = d_mapped;
    prm.frames = frames;
    prm.P = p.P; prm.C = p.C; prm.sides = uint32_t(p.sides);
    prm.chunk = 8;
    prm.numChunks = uint32_t((frames + prm.chunk - 1) / prm.chunk);
    prm.slope = p.d_slope;
    prm.colourTables = p.d_colourTables;
    prm.sc = p.scalars;
    prm.state = d_stateOut; prm.stateIn = d_carry; prm.rgba = d_rgba; prm.lines = d_lines;
    prm.magScale = p.cfg.channel_mode == SGZ_CH_PHASE ? 0.5f : 1.0f;
    if (prm.numChunks > 1) {
        const size_t need = size_t(prm.numChunks) * p.C * p.sides * SGZ_NUM_GRAPHS * p.P;
        if (!p.d_agg || p.aggCap < need || p.aggMapped != d_mapped || p.aggFrames != frames)
            return fail(SGZ_EINVAL, "sgz_stage_decay_emit without a preceding sgz_stage_decay_scan of the same (mapped, frames) -- another K_B call on the plan in between invalidates the kept aggregates");
        prm.agg = p.d_agg;
        if (d_carry) { SGZ_HIP(launchDecayApplyCarry(prm, d_carry, stream)); p.aggMapped = nullptr; }   // (the carry is folded INTO the aggregates: one emit per scan)
    }
    if (!d_rgba && !d_lines && !d_stateOut) return SGZ_OK;
    SGZ_HIP(launchDecayEmit(prm, stream));
    return SGZ_OK;
}

}  // namespace sgz

using namespace sgz;

struct sgz_plan { Plan impl; };

extern "C" {

const char *sgz_last_error(void) { return g_lastError.c_str(); }
int sgz_abi_version(void) { return SGZ_ABI_VERSION; }

int sgz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

sgz_status sgz_set_device(int device)
{
    SGZ_HIP(hipSetDevice(device));
    return SGZ_OK;
}

sgz_status sgz_plan_create(const sgz_spectrum_config *cfg, sgz_plan **out)
{
    if (!cfg || !out) return fail(SGZ_EINVAL, "null argument");
    sgz_plan *pl = new (std::nothrow) sgz_plan();
    if (!pl) return fail(SGZ_ENOMEM, "out of memory");
    std::string err;
    sgz_status st;
    try {
        st = buildPlan(*cfg, pl->impl, err);
    } catch (const std::bad_alloc &) {                                // no exception crosses the ABI
        st = SGZ_ENOMEM; err = "out of memory building the plan tables";
    } catch (const std::exception &e) {
        st = SGZ_EINVAL; err = e.what();
    }
    if (st != SGZ_OK) { delete pl; return fail(st, err); }
    *out = pl;
    return SGZ_OK;
}

void sgz_plan_destroy(sgz_plan *plan) { delete plan; }

sgz_status sgz_plan_upload(sgz_plan *plan)
{
    if (!plan) return fail(SGZ_EINVAL, "null plan");
    std::string err;
    sgz_status st = uploadPlan(plan->impl, err);
    if (st != SGZ_OK && !err.empty()) g_lastError = err;
    return st;
}

uint32_t sgz_plan_transform_size(const sgz_plan *plan) { return plan ? plan->impl.N : 0; }
double sgz_plan_window_scale(const sgz_plan *plan) { return plan ? plan->impl.windowScale : 0.0; }
uint32_t sgz_plan_break_pixel(const sgz_plan *plan) { return plan ? plan->impl.breakPixel : 0; }
sgz_status sgz_plan_set_option(sgz_plan *plan, uint32_t option, uint32_t value)
{
    if (!plan) return fail(SGZ_EINVAL, "null plan");
    Plan &p = plan->impl;
    switch (option) {
    case SGZ_OPT_CHANNEL_SPLIT: p.optChannelSplit = value != 0; return SGZ_OK;
    case SGZ_OPT_FUSED_COLOUR:
        if (value > 1 && value != 4 && value != 8 && value != 16) return fail(SGZ_EINVAL, "SGZ_OPT_FUSED_COLOUR: 0, 1 (= 4 pixels per workgroup), 4, 8 or 16");
        p.optFusedColour = value != 0; p.optFusedPixels = value > 1 ? value : 4u; return SGZ_OK;
    case SGZ_OPT_FETCH_WINDOW: p.optFetchWindow = value != 0; return SGZ_OK;
    case SGZ_OPT_MATRIX_RESONATOR: if (value > 2) return fail(SGZ_EINVAL, "SGZ_OPT_MATRIX_RESONATOR: 0, 1 or 2"); p.optMatrixResonator = int(value); return SGZ_OK;
    case SGZ_OPT_RESONATOR_SLAB: p.optResonatorSlab = value; return SGZ_OK;
    case SGZ_OPT_RESONATOR_SHARD_BOUND: p.optResonatorShardBound = value; return SGZ_OK;
    case SGZ_OPT_PIPELINED: p.optPipelined = value != 0; return SGZ_OK;
    case SGZ_OPT_WIDE_GROUPS: p.optWideGroups = value != 0; return SGZ_OK;
    default: return fail(SGZ_EINVAL, "unknown plan option");
    }
}

uint32_t sgz_plan_path(const sgz_plan *plan)
{
    if (!plan) return SGZ_PATH_GENERIC;
    const Plan &p = plan->impl;
    const uint32_t real = (p.optChannelSplit && (p.realSplit || p.realMono)) ? SGZ_PATH_CHANNEL_SPLIT : 0u;
    if (p.fused) return SGZ_PATH_FUSED | real;
    return (p.halves ? SGZ_PATH_HALVES : SGZ_PATH_GENERIC) | (p.sideMapOk ? SGZ_PATH_SIDE_MAP : 0u) | real;
}
uint32_t sgz_plan_dc_pixels(const sgz_plan *plan, uint32_t *out, uint32_t cap)
{
    if (!plan) return 0;
    const std::vector<uint32_t> &v = plan->impl.dcPixels;
    for (size_t i = 0; out && i < v.size() && i < cap; ++i) out[i] = v[i];
    return uint32_t(v.size());
}

static sgz_status checkReady(sgz_plan *plan);

uint64_t sgz_plan_num_frames(const sgz_plan *plan, size_t nsamples)
{
    if (!plan) return 0;
    const long f = planFrames(plan->impl, nsamples);
    return f > 0 ? uint64_t(f) : 0;
}

sgz_status sgz_plan_get_resonator(const sgz_plan *plan, uint32_t *vectors, float *coeff, float *gain, float *weights)
{
    if (!plan) return fail(SGZ_EINVAL, "null plan");
    const Plan &p = plan->impl;
    if (!isResonator(p)) return fail(SGZ_EINVAL, "not an RSNT plan");
    if (vectors) *vectors = uint32_t(p.resV);
    if (coeff) std::memcpy(coeff, p.resCoeff.data(), p.resCoeff.size() * sizeof(float));
    if (gain) std::memcpy(gain, p.resGain.data(), p.resGain.size() * sizeof(float));
    if (weights) std::memcpy(weights, p.resWeights, size_t(p.resV) * sizeof(float));
    return SGZ_OK;
}

sgz_status sgz_plan_reset_resonator(sgz_plan *plan, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    return resetResonator(plan->impl, reinterpret_cast<hipStream_t>(stream));
}

sgz_status sgz_plan_get_window(const sgz_plan *plan, float *out)
{
    if (!plan || !out) return fail(SGZ_EINVAL, "null argument");
    std::memcpy(out, plan->impl.window.data(), plan->impl.window.size() * sizeof(float));
    return SGZ_OK;
}
sgz_status sgz_plan_get_mapped_frequencies(const sgz_plan *plan, float *out)
{
    if (!plan || !out) return fail(SGZ_EINVAL, "null argument");
    std::memcpy(out, plan->impl.mapped.data(), plan->impl.mapped.size() * sizeof(float));
    return SGZ_OK;
}
sgz_status sgz_plan_get_slope_map(const sgz_plan *plan, float *out)
{
    if (!plan || !out) return fail(SGZ_EINVAL, "null argument");
    std::memcpy(out, plan->impl.slope.data(), plan->impl.slope.size() * sizeof(float));
    return SGZ_OK;
}
sgz_status sgz_plan_get_colour_ratios(const sgz_plan *plan, float *out)
{
    if (!plan || !out) return fail(SGZ_EINVAL, "null argument");
    std::memcpy(out, plan->impl.scalars.ratios, sizeof(plan->impl.scalars.ratios));
    return SGZ_OK;
}
sgz_status sgz_plan_get_colour_table(const sgz_plan *plan, uint32_t pair, float *out)
{
    if (!plan || !out || pair >= plan->impl.C) return fail(SGZ_EINVAL, "bad argument");
    const size_t n = (SGZ_NUM_SPEC_COLOURS + 1) * 3;
    std::memcpy(out, plan->impl.colourTables.data() + size_t(pair) * n, n * sizeof(float));
    return SGZ_OK;
}
void sgz_rotate_hue_rgb8(const uint8_t rgb[3], float amount, uint8_t out[3]) { rotateHueRgb8(rgb, amount, out); }

long sgz_num_frames(size_t nsamples, uint32_t window_size, uint32_t hop)
{
    if (nsamples < window_size || hop == 0) return 0;
    return long((nsamples - window_size) / hop) + 1;
}

static sgz_status checkReady(sgz_plan *plan)
{
    if (!plan) return fail(SGZ_EINVAL, "null plan");
    if (!plan->impl.uploaded) {
        std::string err;
        sgz_status st = uploadPlan(plan->impl, err);
        if (st != SGZ_OK) return fail(st, err);
    }
    return SGZ_OK;
}

sgz_status sgz_spectrogram_render_device(sgz_plan *plan, const float *d_planar, size_t channel_stride,
                                         size_t nsamples, uint8_t *d_rgba, float *d_lines, float *d_state,
                                         void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    Plan &p = plan->impl;
    const long frames = planFrames(p, nsamples);
    if (frames <= 0) return SGZ_SKIPPED_FRAME;      // less than one window: prepareTransform returns false (TransformDSP.inl:45-46)
    if (!d_planar || (!d_rgba && !d_lines)) return fail(SGZ_EINVAL, "null buffer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    st = ensureCap(&p.d_mapped, &p.mappedCap, size_t(frames) * p.C * p.sides * p.P);
    if (st != SGZ_OK) return st;
    // RSNT: a render without a carried decay state is a job of its own -- the resonators start from rest too (TransformPair.h:183);
    // with d_state the caller continues a stream and the resonators (kept in the plan) continue with it
    if (!d_state && (st = resetResonator(p, s)) != SGZ_OK) return st;
    if ((st = runStft(p, d_planar, channel_stride, frames, p.d_mapped, nullptr, nullptr, s, nullptr, /*deferLate=*/true)) != SGZ_OK) return st;
    return runDecayColour(p, p.d_mapped, frames, d_rgba, d_lines, d_state, s);
}

// ---- render queue (sgz.h): `depth` lanes of (plan, stream); submissions go round-robin, every lane is an in-order stream, so a lane's
// scratch is never touched by two renders at once and nothing on the host waits.
struct sgz_render_queue {
    std::vector<sgz_plan *> plans;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;           // per lane: recorded by join() behind the lane's newest render
    std::vector<uint64_t> newest;           // per lane: ticket of that render (0: none yet)
    hipEvent_t inputReady = nullptr;
    uint64_t submitted = 0;
    uint32_t distinct = 0;                  // lanes whose streams were measured to run side by side (the others share a hardware queue)
};

void sgz_render_queue_destroy(sgz_render_queue *q)
{
    if (!q) return;
    for (hipStream_t s : q->streams) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t e : q->done) if (e) (void)hipEventDestroy(e);
    if (q->inputReady) (void)hipEventDestroy(q->inputReady);
    for (sgz_plan *p : q->plans) sgz_plan_destroy(p);
    delete q;
}

// The runtime maps streams onto a few hardware queues (4 unless GPU_MAX_HW_QUEUES says otherwise), in an order that depends on every
// stream the process has made so far -- and two lanes that share a hardware queue run one after the other: the same queue measured
// 25 us per render or 38 (= no overlap at all) depending on how many OTHER streams the process had created (tools/pipeline_depth.py,
// SGZ_DUMMY_STREAMS).  So the lanes' streams are picked by measurement: two kernels that spin for 100 us on the shared 100 MHz clock,
// one per stream -- together they take ~100 us on different hardware queues and ~200 us on the same one.
__global__ void queueProbeSpinKernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static bool streamsRunSideBySide(hipStream_t a, hipStream_t b)
{
    constexpr unsigned long long kTicks = 10000;                // 100 us
    double best = 1e9;
    for (int rep = 0; rep < 2; ++rep) {                         // (the shorter of two tries: a host hiccup can only make a pair look serial)
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(queueProbeSpinKernel, dim3(1), dim3(64), 0, a, kTicks);
        hipLaunchKernelGGL(queueProbeSpinKernel, dim3(1), dim3(64), 0, b, kTicks);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    return best < 170.0;
}

sgz_status sgz_render_queue_create(const sgz_spectrum_config *cfg, uint32_t depth, sgz_render_queue **out)
{
    if (!cfg || !out) return fail(SGZ_EINVAL, "null argument");
    if (depth == 0 || depth > 16) return fail(SGZ_EINVAL, "sgz_render_queue_create: depth 1 .. 16");
    sgz_render_queue *q = new (std::nothrow) sgz_render_queue();
    if (!q) return fail(SGZ_ENOMEM, "out of memory");
    std::vector<hipStream_t> spare;
    auto bail = [&](sgz_status st) { for (hipStream_t s : spare) (void)hipStreamDestroy(s); sgz_render_queue_destroy(q); return st; };
    // streams first: up to 16 candidates, a candidate becomes a lane if it runs side by side with every lane chosen so far
    for (int tries = 0; tries < 16 && q->streams.size() < depth; ++tries) {
        hipStream_t s = nullptr;
        if (hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking); e != hipSuccess) return bail(hipFail(e, "hipStreamCreateWithFlags"));
        bool distinct = true;
        for (hipStream_t chosen : q->streams) distinct = distinct && streamsRunSideBySide(chosen, s);
        if (distinct) q->streams.push_back(s); else spare.push_back(s);
    }
    q->distinct = uint32_t(q->streams.size());
    while (q->streams.size() < depth && !spare.empty()) { q->streams.push_back(spare.back()); spare.pop_back(); }   // more lanes than hardware queues: the rest share
    for (hipStream_t s : spare) (void)hipStreamDestroy(s);
    spare.clear();
    while (q->streams.size() < depth) {
        hipStream_t s = nullptr;
        if (hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking); e != hipSuccess) return bail(hipFail(e, "hipStreamCreateWithFlags"));
        q->streams.push_back(s);
    }
    (void)hipGetLastError();
    for (uint32_t i = 0; i < depth; ++i) {
        sgz_plan *pl = nullptr;
        if (sgz_status st = sgz_plan_create(cfg, &pl); st != SGZ_OK) return bail(st);
        q->plans.push_back(pl);
        // several renders in flight: K_B as few, long workgroups -- it shares the chip with the other lanes' K_A, whose workgroups it
        // displaces CU by CU (tools/pipeline_depth.py: 26.1 -> 25.3 us per render at depth 3; on an idle device the 4-pixel form is faster)
        if (depth >= 2) { pl->impl.optFusedPixels = 16; pl->impl.optPipelined = true; }
        if (sgz_status st = sgz_plan_upload(pl); st != SGZ_OK) return bail(st);
        hipEvent_t ev = nullptr;
        if (hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); e != hipSuccess) return bail(hipFail(e, "hipEventCreateWithFlags"));
        q->done.push_back(ev);
        q->newest.push_back(0);
    }
    if (hipError_t e = hipEventCreateWithFlags(&q->inputReady, hipEventDisableTiming); e != hipSuccess) return bail(hipFail(e, "hipEventCreateWithFlags"));
    *out = q;
    return SGZ_OK;
}

uint32_t sgz_render_queue_distinct_lanes(const sgz_render_queue *q) { return q ? q->distinct : 0; }

sgz_status sgz_render_queue_set_option(sgz_render_queue *q, uint32_t option, uint32_t value)
{
    if (!q) return fail(SGZ_EINVAL, "null queue");
    if (q->submitted) return fail(SGZ_EINVAL, "sgz_render_queue_set_option: before the first submit");
    for (sgz_plan *p : q->plans)
        if (sgz_status st = sgz_plan_set_option(p, option, value); st != SGZ_OK) return st;
    return SGZ_OK;
}

sgz_status sgz_render_queue_submit(sgz_render_queue *q, const float *d_planar, size_t channel_stride, size_t nsamples, uint8_t *d_rgba,
                                   void *after_stream, uint64_t *ticket)
{
    if (!q || !d_planar || !d_rgba) return fail(SGZ_EINVAL, "null argument");
    const size_t lane = size_t(q->submitted % q->plans.size());
    hipStream_t s = q->streams[lane];
    if (after_stream) {
        SGZ_HIP(hipEventRecord(q->inputReady, reinterpret_cast<hipStream_t>(after_stream)));
        SGZ_HIP(hipStreamWaitEvent(s, q->inputReady, 0));
    }
    const sgz_status st = sgz_spectrogram_render_device(q->plans[lane], d_planar, channel_stride, nsamples, d_rgba, nullptr, nullptr, s);
    if (st != SGZ_OK) return st;                               // (SGZ_SKIPPED_FRAME included: nothing was enqueued, no ticket)
    // (no event per submission: a marker behind every render cost ~9 us per render at depth 2-3 -- tools/pipeline_depth.py -- because the
    // lane's next kernel waits for the marker's own completion signal; wait / join look at the lane's stream instead)
    q->newest[lane] = ++q->submitted;
    if (ticket) *ticket = q->submitted;
    return SGZ_OK;
}

sgz_status sgz_render_queue_wait(sgz_render_queue *q, uint64_t ticket)
{
    if (!q) return fail(SGZ_EINVAL, "null queue");
    if (ticket > q->submitted) return fail(SGZ_EINVAL, "sgz_render_queue_wait: no such ticket");
    // a lane's stream is in order: submission `ticket` has finished once its lane has drained (which may include later submissions
    // of the same lane: the wait is for at least the ticket)
    for (size_t lane = 0; lane < q->plans.size(); ++lane) {
        if (q->newest[lane] == 0) continue;
        if (ticket == 0 || size_t((ticket - 1) % q->plans.size()) == lane) SGZ_HIP(hipStreamSynchronize(q->streams[lane]));
    }
    return SGZ_OK;
}

sgz_status sgz_render_queue_join(sgz_render_queue *q, void *stream)
{
    if (!q) return fail(SGZ_EINVAL, "null queue");
    for (size_t lane = 0; lane < q->plans.size(); ++lane)
        if (q->newest[lane]) {
            SGZ_HIP(hipEventRecord(q->done[lane], q->streams[lane]));
            SGZ_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), q->done[lane], 0));
        }
    return SGZ_OK;
}

// Host buffers in, host buffers out, on a plan the caller keeps: the constant block is built and uploaded once, the device buffers and
// the stream live in the plan and only grow, so a second render of the same shape allocates nothing.
sgz_status sgz_spectrogram_render_host(sgz_plan *plan, const float *const *planar, uint32_t num_channels, size_t nsamples,
                                       uint8_t *rgba_out, float *lines_out, sgz_timing *timing)
{
    if (!plan || !planar || !rgba_out) return fail(SGZ_EINVAL, "null argument");
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    Plan &p = plan->impl;
    if (num_channels != 2 * p.C) return fail(SGZ_EINVAL, "num_channels must equal 2*num_pairs (SpectrumDSP.cpp:65-72)");
    const long frames = planFrames(p, nsamples);
    if (frames <= 0) { if (timing) *timing = sgz_timing{}; return SGZ_SKIPPED_FRAME; }
    if (!p.hostStream) {
        hipStream_t ns = nullptr;
        SGZ_HIP(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        p.hostStream = ns;
        for (void *&e : p.hostEv) { hipEvent_t ne = nullptr; SGZ_HIP(hipEventCreate(&ne)); e = ne; }
    }
    hipStream_t s = static_cast<hipStream_t>(p.hostStream);
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) ev[i] = static_cast<hipEvent_t>(p.hostEv[i]);
    const size_t stride = (nsamples + 63) & ~size_t(63);                    // 256-byte rows
    const size_t linesN = size_t(frames) * p.C * SGZ_NUM_GRAPHS * p.P * 2;
    if ((st = ensureCap(&p.d_hostAudio, &p.hostAudioCap, size_t(num_channels) * stride)) != SGZ_OK) return st;
    if ((st = ensureCap(&p.d_hostRgba, &p.hostRgbaCap, (size_t(frames) * p.P * 4 + 3) / 4)) != SGZ_OK) return st;
    if (lines_out && (st = ensureCap(&p.d_hostLines, &p.hostLinesCap, linesN)) != SGZ_OK) return st;
    SGZ_HIP(hipEventRecord(ev[0], s));
    for (uint32_t c = 0; c < num_channels; ++c)
        SGZ_HIP(hipMemcpyAsync(p.d_hostAudio + size_t(c) * stride, planar[c], nsamples * sizeof(float), hipMemcpyHostToDevice, s));
    SGZ_HIP(hipEventRecord(ev[1], s));
    uint8_t *d_rgba = reinterpret_cast<uint8_t *>(p.d_hostRgba);
    st = sgz_spectrogram_render_device(plan, p.d_hostAudio, stride, nsamples, d_rgba, lines_out ? p.d_hostLines : nullptr, nullptr, s);
    if (st != SGZ_OK) return st;
    SGZ_HIP(hipEventRecord(ev[2], s));
    SGZ_HIP(hipMemcpyAsync(rgba_out, d_rgba, size_t(frames) * p.P * 4, hipMemcpyDeviceToHost, s));
    if (lines_out) SGZ_HIP(hipMemcpyAsync(lines_out, p.d_hostLines, linesN * sizeof(float), hipMemcpyDeviceToHost, s));
    SGZ_HIP(hipEventRecord(ev[3], s));
    SGZ_HIP(hipStreamSynchronize(s));
    if (timing) {
        float a = 0, b = 0, c = 0;
        (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]);
        (void)hipEventElapsedTime(&c, ev[2], ev[3]);
        timing->h2d_ms = a; timing->kernel_ms = b; timing->d2h_ms = c; timing->frames = uint64_t(frames);
    }
    return SGZ_OK;
}

// One-shot convenience: builds a plan for cfg, renders, destroys it.  Callers that render more than once keep the plan
// (sgz_plan_create + sgz_spectrogram_render_host): the constant block (fp64 window design, pixel records) costs more than the render.
sgz_status sgz_spectrogram_render(const sgz_spectrum_config *cfg, const float *const *planar, uint32_t num_channels,
                                  size_t nsamples, uint8_t *rgba_out, float *lines_out, sgz_timing *timing)
{
    if (!cfg || !planar || !rgba_out) return fail(SGZ_EINVAL, "null argument");
    if (num_channels != 2 * cfg->num_pairs) return fail(SGZ_EINVAL, "num_channels must equal 2*num_pairs (SpectrumDSP.cpp:65-72)");
    sgz_plan *plan = nullptr;
    sgz_status st = sgz_plan_create(cfg, &plan);
    if (st != SGZ_OK) return st;
    st = sgz_spectrogram_render_host(plan, planar, num_channels, nsamples, rgba_out, lines_out, timing);
    sgz_plan_destroy(plan);
    return st;
}

// Device memory another API can import: allocated here, exported as a dma-buf fd (whole pages).  For the vertex / image hand-off to a
// GL or Vulkan context on the display GPU (an MI355X has no graphics engine): glImportMemoryFdEXT + glNamedBufferStorageMemEXT.
sgz_status sgz_export_alloc(size_t bytes, void **d_ptr, size_t *allocated, int *dmabuf_fd)
{
    if (!bytes || !d_ptr) return fail(SGZ_EINVAL, "bad argument");
    // A dma-buf is a whole buffer object.  The HSA runtime carves allocations below 2 MiB out of shared 2 MiB blocks, and an importer of
    // such an fd sees the BLOCK from its start, not the allocation (measured: another process read the neighbouring tables through
    // the fd of a 52 KB image).  Exported memory is therefore allocated in whole 2 MiB blocks, which get a buffer object of their own.
    const size_t rounded = (bytes + kExportGranule - 1) & ~(kExportGranule - 1);
    void *p = nullptr;
    SGZ_HIP(hipMalloc(&p, rounded));
    int fd = -1;
    if (dmabuf_fd) {
        const hipError_t e = hipMemGetHandleForAddressRange(&fd, p, rounded, hipMemRangeHandleTypeDmaBufFd, 0);
        if (e != hipSuccess) { (void)hipFree(p); return hipFail(e, "hipMemGetHandleForAddressRange (dma-buf export)"); }
        *dmabuf_fd = fd;
    }
    *d_ptr = p;
    if (allocated) *allocated = rounded;
    return SGZ_OK;
}
void sgz_export_free(void *d_ptr) { if (d_ptr) (void)hipFree(d_ptr); }

sgz_status sgz_stage_bins(sgz_plan *plan, const float *d_planar, size_t channel_stride, size_t nsamples,
                          float *d_bins, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    Plan &p = plan->impl;
    const long frames = planFrames(p, nsamples);
    return runStft(p, d_planar, channel_stride, frames, nullptr, d_bins, nullptr, reinterpret_cast<hipStream_t>(stream));
}

sgz_status sgz_stage_mapped(sgz_plan *plan, const float *d_planar, size_t channel_stride, size_t nsamples,
                            float *d_mapped, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    Plan &p = plan->impl;
    const long frames = planFrames(p, nsamples);
    if ((st = resetResonator(p, reinterpret_cast<hipStream_t>(stream))) != SGZ_OK) return st;     // RSNT: a stage call starts from rest
    return runStft(p, d_planar, channel_stride, frames, d_mapped, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream));
}

sgz_status sgz_stage_mapped_dominant(sgz_plan *plan, const float *d_planar, size_t channel_stride, size_t nsamples,
                                     float *d_mapped, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    Plan &p = plan->impl;
    const long frames = planFrames(p, nsamples);
    if ((st = resetResonator(p, reinterpret_cast<hipStream_t>(stream))) != SGZ_OK) return st;
    st = runStft(p, d_planar, channel_stride, frames, d_mapped, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream), nullptr, /*deferLate=*/true);
    p.lateDeferred = nullptr;                        // (nobody completes them: see sgz.h)
    return st;
}

sgz_status sgz_stage_map_from_bins(sgz_plan *plan, const float *d_bins, size_t frames, float *d_mapped, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    return runStft(plan->impl, nullptr, 0, long(frames), d_mapped, nullptr, d_bins, reinterpret_cast<hipStream_t>(stream));
}

sgz_status sgz_stage_decay_colour(sgz_plan *plan, const float *d_mapped, size_t frames, uint8_t *d_rgba,
                                  float *d_lines, float *d_state, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    return runDecayColour(plan->impl, d_mapped, long(frames), d_rgba, d_lines, d_state, reinterpret_cast<hipStream_t>(stream));
}

sgz_status sgz_stage_decay_scan(sgz_plan *plan, const float *d_mapped, size_t frames, float *d_end_state, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    Plan &p = plan->impl;
    if (!d_mapped || !d_end_state) return fail(SGZ_EINVAL, "null buffer");
    // Phase mode: the MAGNITUDE half of the state (what the image is coloured from, SpectrumDSP.cpp:123) is the same peak decay and
    // folds exactly; the cancellation smoother (a linear recurrence: no exact fold) is not carried -- sgz_stage_decay_emit then
    // renders the image only
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    SGZ_HIP(hipMemsetAsync(d_end_state, 0, size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2 * sizeof(float), s));
    return runDecayColour(p, d_mapped, long(frames), nullptr, nullptr, d_end_state, s, /*magnitudeOnly=*/true);     // scans + state-only emit of the last chunk
}

sgz_status sgz_stage_decay_emit(sgz_plan *plan, const float *d_mapped, size_t frames, const float *d_carry, uint8_t *d_rgba,
                                float *d_lines, float *d_state_out, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    if (!d_mapped) return fail(SGZ_EINVAL, "null buffer");
    if (plan->impl.cfg.channel_mode == SGZ_CH_PHASE && (d_lines || d_state_out))
        return fail(SGZ_EUNSUPPORTED, "Phase mode: only the image can be rendered from a folded carry (the cancellation smoother is a linear recurrence without an exact fold)");
    return runDecayEmitWithCarry(plan->impl, d_mapped, long(frames), d_carry, d_rgba, d_lines, d_state_out, reinterpret_cast<hipStream_t>(stream));
}

sgz_status sgz_stage_logf(const float *d_x, float *d_y, size_t n, void *stream)
{
    if (!d_x || !d_y) return fail(SGZ_EINVAL, "null buffer");
    SGZ_HIP(launchLogf(d_x, d_y, n, reinterpret_cast<hipStream_t>(stream)));
    return SGZ_OK;
}

sgz_status sgz_stage_finish_pixel(const float *d_x, float *d_y, size_t n, void *stream)
{
    if (!d_x || !d_y) return fail(SGZ_EINVAL, "null buffer");
    SGZ_HIP(launchFinishPixel(d_x, d_y, n, reinterpret_cast<hipStream_t>(stream)));
    return SGZ_OK;
}

#ifdef SGZ_DEBUG
/* debug hooks of a -DSGZ_DEBUG build only (not in sgz.h): phase ablation bits, and per-phase shader clocks of one workgroup of K_A;
 * d_clocks: DEVICE uint64[16 slots x 16 waves] */
void sgz_debug_set_ablate(uint32_t bits) { g_ablate = bits; }

sgz_status sgz_debug_phase_clocks(sgz_plan *plan, const float *d_planar, size_t channel_stride, size_t nsamples,
                                  float *d_mapped, unsigned long long *d_clocks, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    Plan &p = plan->impl;
    const long frames = planFrames(p, nsamples);
    return runStft(p, d_planar, channel_stride, frames, d_mapped, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream), d_clocks);
}
#endif

sgz_status sgz_decay_fold_carry(sgz_plan *plan, const float *d_aggs, const int64_t *frames_per_rank, uint32_t world,
                                uint32_t rank, float *d_carry, void *stream)
{
    sgz_status st = checkReady(plan);
    if (st != SGZ_OK) return st;
    if (!d_aggs || !frames_per_rank || !d_carry || rank >= world || world > 64) return fail(SGZ_EINVAL, "bad argument");
    Plan &p = plan->impl;
    long long fr[64];                                        // (Phase: the magnitude halves fold exactly; see sgz_stage_decay_scan)
    for (uint32_t q = 0; q < world; ++q) fr[q] = frames_per_rank[q];
    const size_t perRank = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    SGZ_HIP(launchDecayFold(d_aggs, fr, world, rank, perRank, p.P, p.scalars, d_carry, reinterpret_cast<hipStream_t>(stream)));
    return SGZ_OK;
}

}  // extern "C"
