// scope_stream.hip -- the Oscilloscope's real-time handle (sgz_scope_*): the whole audio-thread state machine lives in HBM and
// runs as ONE kernel launch per audio block, so sgz_scope_push only stages the block and enqueues -- no device -> host round trip
// for the trigger list, no host copy of the rings.  gfx950 only.
//
// Replaces, per onStreamAudio callback (Source/Oscilloscope/Oscilloscope.h:293 -> StreamState::audioEntryPoint,
// OscilloscopeDSP.inl:401-424):
//   TriggeringProcessor::update                          StreamPreprocessing.h:55-76
//   preAnalyseAudio -> ZeroCrossingProcessor::process    OscilloscopeDSP.inl:311-399, StreamPreprocessing.h:315-349
//   TriggeringProcessor::processMutating                 StreamPreprocessing.h:79-206  (trigger -> window selection)
//   ChannelData::swapBuffers                             ChannelData.h:147-161         (back -> front ring)
//   StreamState::audioProcessing: RMS envelope + writes  OscilloscopeDSP.inl:520-585, :676-697
// and on the render thread: Oscilloscope::runPeakFilter (OscilloscopeDSP.inl:713-886, every OscChannels mode) and drawWavePlot's
// Linear and Lanczos branches (OscilloscopeRendering.cpp:551-649, :707-741, :790-891) -> (x, y, z) + RGBA8 per vertex.
//
// Layout in HBM.  front: [channels][size] rings, size = ceil(window + 1) (ChannelData::resizeAudioStorage), one write cursor for
// all channels -- exactly the memory the reference's CLIFOStream proxy views expose (begin() / cursorPosition()), so memory-order
// semantics (runPeakFilter's dropped tail) carry over.  back: [channels][backCap] indexed by ABSOLUTE sample number mod backCap
// (backCap = power of two >= size): "cursor - bufferedSamples" of the reference is absolute index  written - bufferedSamples.
// A swap's source range may straddle the history and the block being ingested; the copy reads either.
//
// The block kernel (one 1024-thread workgroup -- a block is a few thousand samples):
//   A  trigger detection as two prefix-max scans + ordered compaction (the automaton's closed form, scope_vector.hip K10) into
//      the device-resident peaks queue;
//   B  thread 0 replays update() + processMutating's integer automaton (same uint64 / double conversions as the reference) and
//      emits the list of swaps (source sample, length) and the extent of the last audioProcessing call;
//   C  all threads execute the swaps in order into the front rings;
//   D  the block is appended to the back rings;
//   E  RMS envelope: one lane per recurrence, sequential fp32 (rounding order is the reference's).
//   F  (colour_by_frequency) the per-sample colours of audioProcessing (:445-647) for the whole block, before C so that swaps can
//      take them along: F1/F2 the 3-band Linkwitz-Riley tree as two passes of cascaded biquads (one lane per channel and branch,
//      sequential in time), F3 the twelve band-energy smoothers of a channel pair (left / right / mid / side x 3 bands, one lane
//      each), F4 accumulateColour for every (sample, signal) in parallel.  Colour rings sit beside the audio rings (colourData /
//      auxColourData, ChannelData.h:58-66) and move with them (swapBuffers, ChannelData.h:155-159).
//
// Spectral triggering (sgz_scope_analyse -> scopeSpectralKernel, OscilloscopeDSP.inl:62-308): audio goes straight into the front ring
// like TriggeringMode::None; once per rendered frame one workgroup transforms the newest 8192 samples in LDS (fp64), picks the
// fundamental, runs the median of 8 and the Goertzel phase, and leaves cycleSamples / sampleOffset for the vertex kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>

#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "rt_common.hpp"

#pragma clang fp contract(off)

using namespace sgz;

namespace sgz {
// scope_vector.hip: Lanczos / linear vertex kernels on a ring whose cursor lives in device memory
bool launchScopeLanczosPair(const sgz_scope_view &view, uint32_t triggerMode, uint32_t interpolation, const float *const ringA[2],
                            const float *const ringB[2], const uint32_t evalMode[2], uint32_t size, uint32_t cap, const uint32_t *d_cursor,
                            double cycleSamples, double sampleOffset, long long transport, const uint32_t key[2], const uint32_t *const colRing[2],
                            float *const d_xyz[2], uint32_t *const d_rgba[2], size_t capacity, size_t *points, hipStream_t stream, hipError_t *err);
hipError_t launchScopeVertices(const sgz_scope_view &view, uint32_t triggerMode, uint32_t interpolation, const float *ringA,
                               const float *ringB, uint32_t evalMode, uint32_t size, uint32_t cap, const uint32_t *d_cursor,
                               double cycleSamples, double sampleOffset, long long transport, uint32_t rgba, const uint32_t *colRing, float *d_xyz,
                               uint32_t *d_rgba, size_t capacity, size_t *points, hipStream_t stream);
size_t scopeVertexCount(const sgz_scope_view &view, uint32_t interpolation, uint32_t triggerMode, double cycleSamples);
}

namespace {

constexpr uint32_t kPeakCap = 1u << 16;       // pending triggers (std::queue<std::uint64_t> peaks)
constexpr uint32_t kMaxSwaps = kPeakCap + 8;  // one swap pops one trigger
constexpr uint32_t kMaxCh = 64;

struct Swap { unsigned long long src; unsigned int len; unsigned int pad; };

struct ScopeDev {
    // TriggeringProcessor (StreamPreprocessing.h:210-226)
    double threshold, windowSize, state, hysteresis;
    int windowChanged, isPeakHold, isWorkingOnPeak, pad0;
    unsigned long long crossOrigin, oldPeak, currentPeak, bufferedSamples, frontOrigin, steadyClock;
    unsigned long long playhead;              // ctx.getPlayhead().getSteadyClock(): samples delivered so far
    unsigned int qHead, qCount;               // peaks queue (ring of kPeakCap)
    // rings
    unsigned long long written;               // samples ever appended to the back rings
    unsigned int frontCursor, pad1;
    // envelope
    float envelope[kMaxCh];                   // filterStates.channels[c].envelope
    double envelopeGain;                      // StreamState::envelopeGain
    double autoGain;                          // state.autoGain after runPeakFilter
    unsigned long long swaps, droppedPeaks;
};

// frequency colouring state: filterStates.channels[c].{network, smoothFilters, auxSmoothFilter} (ChannelData.h:68-80)
struct ColourDev {
    float z[kMaxCh][8][2];                    // biquad states: lp1 a, b; hp1 a, b; lp2 a, b; hp2 a, b (transposed direct form II)
    float smooth[kMaxCh][3], aux[kMaxCh][3];
};
struct ColourParams {
    ColourDev *st;
    float lp1[5], hp1[5], lp2[5], hp2[5];     // b0 b1 b2 a1 a2 (Crossover::Coefficients::design, UNVERIFIED vs cpl: see oracle/scope_spectral.c)
    float pole, blend;                        // channelData.smoothFilterPole; 1 - frequencyColouringBlend
    float band[3][3];                         // low / mid / high colour
    float *bands;                             // scratch [channels][4][maxBlock]: low, mid, high, rest
    float *sm;                                // scratch [channels / 2 * 12][maxBlock]: smoothed band energies
    uint32_t *block;                          // [2 channels][maxBlock] colours of the block: colourData planes, then auxColourData
    uint32_t *front, *back;                   // [2 channels][size] / [2 channels][backCap]
    uint32_t maxBlock;
    uint32_t keys[kMaxCh];                    // defaultKey per channel, RGBA8
};

struct PeakParams {           // Oscilloscope::runPeakFilter (scopePeakBody below)
    ScopeDev *st;
    const float *front; uint32_t size, channels, mode, lanes;
    double coeff;
    uint32_t len;       // Spectral: the reference's ring of the moment (the newest len samples, taken as memory order); else 0
};

struct IngestParams {
    ScopeDev *st;
    unsigned long long *peaks;                // [kPeakCap]
    Swap *swapList;                           // [kMaxSwaps]
    const float *batch;                       // the staged blocks, back to back: block b = [channels][blockLen[b]] at batch + blockOff[b]
    const float *batchHost; uint32_t batchFloats;   // the pinned slot they are fetched from first (rt_common.hpp batchFetch), or null
    uint32_t numBlocks, channels;
    uint32_t blockOff[BatchRing::kMaxBlocks], blockLen[BatchRing::kMaxBlocks];
    float *front; uint32_t size;              // [channels][size]
    float *back; uint32_t backCap;            // [channels][backCap], power of two
    uint32_t triggerMode, oscMode, envMode;
    uint32_t trigSeparate, trigPair;
    float envelopeCoeff;
    uint32_t colours;                         // colour_by_frequency
    uint32_t doPeak; PeakParams peak;         // the render thread's peak filter behind the batch's last block (sgz_scope_peak_filter found a batch waiting: one launch)
};

// A lane that walks a block sequentially (a recurrence) must not wait a memory round trip per sample: the samples come in batches of
// 16 independent loads, the next batch in flight while the current one is consumed.  load(i) -> float, step(i, v).
template <typename Load, typename Step>
__device__ __forceinline__ void walkSequential(uint32_t n, Load load, Step step)
{
    constexpr uint32_t B = 16;
    const uint32_t full = n & ~(B - 1);                      // whole batches: straight-line code, no per-sample predicate
    if (full) {
        float cur[B], nxt[B];
#pragma unroll
        for (uint32_t j = 0; j < B; ++j) cur[j] = load(j);
        for (uint32_t i0 = 0; i0 < full; i0 += B) {
            const uint32_t in = i0 + B < full ? i0 + B : i0;  // (the last batch reloads itself: uniform code)
#pragma unroll
            for (uint32_t j = 0; j < B; ++j) nxt[j] = load(in + j);
#pragma unroll
            for (uint32_t j = 0; j < B; ++j) step(i0 + j, cur[j]);
#pragma unroll
            for (uint32_t j = 0; j < B; ++j) cur[j] = nxt[j];
        }
    }
    for (uint32_t i = full; i < n; ++i) step(i, load(i));     // < 16 samples
}

__device__ __forceinline__ float biquadStep(const float *c, float &z0, float &z1, float x)
{
    const float y = c[0] * x + z0;
    z0 = (c[1] * x - c[3] * y) + z1;
    z1 = c[2] * x - c[4] * y;
    return y;
}
// static_cast<uint8_t>(float) as the x86 build behaves (see oracle/scope_spectral.c to_u8)
__device__ __forceinline__ uint32_t toU8(float v) { return (!(v > -1.0f) || !(v < 256.0f)) ? 0u : uint32_t(int(v)) & 255u; }
// accumulateColour (OscilloscopeDSP.inl:472-497) + PixelType::lerp(key, blend)
__device__ __forceinline__ uint32_t accumulateColour(const float st[3], const float band[3][3], uint32_t key, float blend)
{
    float red = 0.f, green = 0.f, blue = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        red += st[i] * band[i][0];
        green += st[i] * band[i][1];
        blue += st[i] * band[i][2];
    }
    const float invMax = 255.0f / fmaxf(red, fmaxf(blue, green));
    const uint32_t ret[4] = {toU8(red * invMax), toU8(green * invMax), toU8(blue * invMax), 255u};
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = float(ret[k]), b = float((key >> (8 * k)) & 255u);
        out |= toU8(a + (b - a) * blend) << (8 * k);
    }
    return out;
}

// two exclusive prefix-max scans at once on 32-bit values (the zero-crossing phase's "last arm" / "last threshold crossing": sample
// indices of one block and small negative sentinels) -- one pass of shuffles, one pair of barriers.  (As two 64-bit scans they were
// 3.7 us of the phase's 6.6: every 64-bit shuffle is two permutes, every scan two barriers of sixteen waves.)
__device__ __forceinline__ void blockExclusiveMax2(int va, int initA, int vb, int initB, int *lds /*[32]*/, int &inA, int &inB, int &totA, int &totB)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    int ia = va, ib = vb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ua = __shfl_up(ia, o), ub = __shfl_up(ib, o);
        if (lane >= o) { ia = ua > ia ? ua : ia; ib = ub > ib ? ub : ib; }
    }
    if (lane == 63) { lds[wave] = ia; lds[16 + wave] = ib; }
    __syncthreads();
    int baseA = initA, baseB = initB, ta = initA, tb = initB;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int pa = lds[w], pb = lds[16 + w];
        if (w < wave) { baseA = pa > baseA ? pa : baseA; baseB = pb > baseB ? pb : baseB; }
        if (w < waves) { ta = pa > ta ? pa : ta; tb = pb > tb ? pb : tb; }
    }
    int prevA = __shfl_up(ia, 1), prevB = __shfl_up(ib, 1);
    if (lane == 0) { prevA = INT_MIN; prevB = INT_MIN; }
    __syncthreads();
    totA = ta; totB = tb;
    inA = prevA > baseA ? prevA : baseA;
    inB = prevB > baseB ? prevB : baseB;
}
__device__ __forceinline__ unsigned int blockExclusiveSum(unsigned int v, unsigned int *lds, unsigned int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    unsigned int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    unsigned int base = 0, tot = 0;
    for (int w = 0; w < waves; ++w) { if (w < wave) base += lds[w]; tot += lds[w]; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// inclusive prefix sum over the workgroup (all threads call it); lds: 16 words of T
template <typename T>
__device__ __forceinline__ T blockInclusiveSum(T v, T *lds, T *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    T base = 0, tot = 0;
    for (int w = 0; w < waves; ++w) { if (w < wave) base += lds[w]; tot += lds[w]; }
    __syncthreads();
    *total = tot;
    return base + inc;
}

__device__ __forceinline__ double trigSample(uint32_t mode, const float *a, const float *b, uint32_t i)
{
    if (mode == SGZ_OSC_MID) return double(0.5f * (a[i] + b[i]));          // OscilloscopeDSP.inl:371-376
    if (mode == SGZ_OSC_SIDE) return double(0.5f * (a[i] - b[i]));         // :377-382
    return double(a[i]);
}

__device__ __forceinline__ unsigned long long minU64(unsigned long long a, unsigned long long b) { return a < b ? a : b; }

#ifdef SGZ_DEBUG
__device__ unsigned long long g_ingestClk[8];
#define ICLK(k) do { if (threadIdx.x == 0) g_ingestClk[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ICLK(k) do { } while (0)
#endif
// (the colour parameters -- ~100 words with the per-channel keys -- come through a pointer: as a by-value argument they stayed live in
// scalar registers across the loop over the batch's blocks and pushed four vector registers into scratch)
__device__ void scopePeakBody(const PeakParams &prm);

__global__ void __launch_bounds__(1024) scopeIngestKernel(const IngestParams prm, const ColourParams *colp)
{
    const ColourParams &col = *colp;
    __shared__ int sScan2[32];
    __shared__ unsigned int sSum[16];
    __shared__ unsigned int sNumSwaps, sLastStart, sLastLen, sCursor0;
    __shared__ unsigned long long sWritten0;
    // The stream state lives in LDS for the length of the kernel (one coalesced read, one write-back): the phases below touch its fields
    // some thirty times, mostly from one lane and each time behind the last (the kernels of the render thread read the copy in HBM,
    // between launches)
    __shared__ ScopeDev sState;
    const uint32_t C = prm.channels;
    const int tid = threadIdx.x, T = blockDim.x;
    static_assert(sizeof(ScopeDev) % 4 == 0, "copied as words");
    for (uint32_t w = tid; w < sizeof(ScopeDev) / 4; w += T) reinterpret_cast<uint32_t *>(&sState)[w] = reinterpret_cast<const uint32_t *>(prm.st)[w];
    __syncthreads();
    ScopeDev *st = &sState;
    constexpr unsigned int kStage = 64;             // triggers / swaps staged in LDS
    constexpr unsigned int kFlat = 1024;            // swaps of a callback the one-pass copy takes (one per thread; more: one after the other)
    __shared__ unsigned long long sPeaks[kStage];
    __shared__ Swap sSwaps[kStage];
    __shared__ unsigned long long sFlatSrc[kFlat];
    __shared__ uint32_t sFlatDst[kFlat], sFlatEnd[kFlat], sFlatTotal, sCursorEnd;
    __shared__ unsigned long long sScan64[16];
    // One launch ingests every block that was waiting (sgz_scope_push only stages; whoever needs the state -- the render thread's calls, a
    // full batch, sgz_scope_flush -- submits): the blocks go through the reference's per-callback state machine ONE AFTER THE OTHER, with
    // their boundaries where the host put them (audioEntryPoint runs once per onStreamAudio: update(), the detector, processMutating's
    // window selection all see callback extents), the stream state staying in LDS in between.
    // (the block table goes through LDS: a run-time subscript into the by-value argument struct would move the struct to scratch)
    __shared__ uint32_t sBlockOff[BatchRing::kMaxBlocks], sBlockLen[BatchRing::kMaxBlocks];
#pragma unroll
    for (uint32_t b = 0; b < BatchRing::kMaxBlocks; ++b)
        if (tid == int(b)) { sBlockOff[b] = prm.blockOff[b]; sBlockLen[b] = prm.blockLen[b]; }
    __syncthreads();
    batchFetch(prm.batchHost, const_cast<float *>(prm.batch), prm.batchFloats, tid, T);
    // ---- A for the WHOLE batch.  The zero-crossing detector (phase A below) is a scan over the samples whose state (armed, last
    // threshold crossing, previous sample) does not depend on what processMutating does with the triggers, and a trigger's slot in the
    // queue's ring is head + count + (triggers before it) whatever has been popped meanwhile: so the detector runs ONCE over the
    // concatenation of the batch's blocks (three passes and two block-wide scans instead of that per block: 4 us per block saved),
    // writes every trigger where the per-callback walk would, and each block's processMutating then sees the queue grow by its own
    // block's triggers only (sFires).  Taken when no trigger can be dropped (count + all fires <= capacity: then no callback of the
    // sequential walk drops one either) and no window change is pending (update() edits the queue); otherwise block by block as before.
    __shared__ uint32_t sBlockStart[BatchRing::kMaxBlocks + 1];
    __shared__ unsigned int sFires[BatchRing::kMaxBlocks];
    __shared__ int sBatchedA;
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t b = 0; b < BatchRing::kMaxBlocks; ++b) { sBlockStart[b] = acc; acc += b < prm.numBlocks ? sBlockLen[b] : 0u; }
        sBlockStart[BatchRing::kMaxBlocks] = acc;
        sBatchedA = 0;
    }
    if (tid < int(BatchRing::kMaxBlocks)) sFires[tid] = 0;
    __syncthreads();
    if (prm.triggerMode == 4u && C >= 2 && prm.numBlocks > 1 && !st->windowChanged) {       // (uniform)
        uint32_t localMode = prm.oscMode, pair = prm.trigPair;
        if (localMode == SGZ_OSC_MIDSIDE) { localMode = SGZ_OSC_MID; pair = prm.trigSeparate & ~1u; }     // :340-352
        const uint32_t N = sBlockStart[prm.numBlocks];
        // sample i of the concatenation; b: the block it is in (kept by the caller: a thread walks forward)
        auto sampleAt = [&](uint32_t i, uint32_t &b) -> double {
            while (sBlockStart[b + 1] <= i) ++b;
            const uint32_t nb = sBlockLen[b], j = i - sBlockStart[b];
            const float *base = prm.batch + sBlockOff[b];
            const float *pa, *pb;
            if (localMode == SGZ_OSC_RIGHT) pa = pb = base + size_t(pair + 1) * nb;
            else if (localMode == SGZ_OSC_LEFT) pa = pb = base + size_t(pair) * nb;
            else if (localMode == SGZ_OSC_SEPARATE) pa = pb = base + size_t(prm.trigSeparate) * nb;
            else { pa = base + size_t(pair) * nb; pb = pa + nb; }
            return trigSample(localMode, pa, pb, j);
        };
        const double threshold = st->threshold, prevState = st->state;
        const int armedIn = st->isPeakHold;
        const unsigned long long originIn = st->crossOrigin, playhead0 = st->playhead;
        const uint32_t seg = (N + T - 1) / T;
        const uint32_t i0 = min(N, uint32_t(tid) * seg), i1 = min(N, i0 + seg);
        uint32_t bStart = 0;                                               // the block of sample i0 - 1 (or of i0)
        double before = prevState;                                         // sample i0 - 1
        if (i0 > 0 && i0 <= N) before = sampleAt(i0 - 1, bStart);
        int segArm = INT_MIN, segThr = INT_MIN;
        {
            uint32_t b = bStart; double prev = before;
            for (uint32_t i = i0; i < i1; ++i) {
                const double s = sampleAt(i, b);
                if (s > 0 && prev < 0) segArm = int(i);
                if (s > threshold) segThr = int(i);
                prev = s;
            }
        }
        int totArm, totThr, inArm, inThr;
        blockExclusiveMax2(segArm, armedIn ? -1 : -3, segThr, -2, sScan2, inArm, inThr, totArm, totThr);
        unsigned int fires = 0;
        {
            uint32_t b = bStart; double prev = before;
            int la = inArm, lt = inThr;
            for (uint32_t i = i0; i < i1; ++i) {
                const double s = sampleAt(i, b);
                if (s > 0 && prev < 0) la = int(i);
                if (s > threshold) { if (la > lt) { ++fires; atomicAdd(&sFires[b], 1u); } lt = int(i); }
                prev = s;
            }
        }
        unsigned int totalFires;
        unsigned int pos = blockExclusiveSum(fires, sSum, &totalFires);
        const unsigned int q0 = st->qHead, qc0 = st->qCount;
        if (qc0 + totalFires <= kPeakCap) {                                // (uniform) nothing can be dropped
            uint32_t b = bStart; double prev = before;
            int la = inArm, lt = inThr;
            for (uint32_t i = i0; i < i1; ++i) {
                const double s = sampleAt(i, b);
                if (s > 0 && prev < 0) la = int(i);
                if (s > threshold) {
                    if (la > lt) {
                        prm.peaks[(q0 + qc0 + pos) % kPeakCap] = (la == -1) ? originIn : playhead0 + (unsigned long long)la;
                        ++pos;
                    }
                    lt = int(i);
                }
                prev = s;
            }
            __syncthreads();
            if (tid == 0) {
                st->isPeakHold = totArm > totThr ? 1 : 0;
                if (totArm >= 0) st->crossOrigin = playhead0 + (unsigned long long)totArm;
                uint32_t bl = prm.numBlocks - 1;
                st->state = sampleAt(N - 1, bl);
                sBatchedA = 1;
            }
        }
        __syncthreads();
    }
    const bool batchedA = sBatchedA != 0;
    for (uint32_t blockIndex = 0; blockIndex < prm.numBlocks; ++blockIndex) {
    const float *const blk = prm.batch + sBlockOff[blockIndex];
    const uint32_t n = sBlockLen[blockIndex];
    const unsigned long long playhead = st->playhead;
    ICLK(0);

    // ---- update(), StreamPreprocessing.h:55-76 (thread 0; the queue edits must precede phase A's appends)
    if (tid == 0) {
        st->steadyClock = playhead;
        if (st->windowChanged) {
            st->windowChanged = 0;
            if (st->isWorkingOnPeak && st->qCount) { st->qHead = (st->qHead + 1) % kPeakCap; st->qCount--; }
            while (st->qCount && prm.peaks[st->qHead] < playhead) { st->qHead = (st->qHead + 1) % kPeakCap; st->qCount--; }
            st->bufferedSamples = st->currentPeak = st->oldPeak = 0;
            st->frontOrigin = playhead;
            st->isWorkingOnPeak = 0;
        }
    }
    __syncthreads();

    const bool hold = prm.triggerMode == 4u || prm.triggerMode == 3u;      // ZeroCrossing, EnvelopeHold: detector -> processMutating
    // ---- A': PeakHoldProcessor over the block (StreamPreprocessing.h:270-313): an envelope follower whose every step depends on the
    // last one's branch -- one lane walks the block (a few microseconds per 512 samples; the triggers it finds are rare)
    if (prm.triggerMode == 3u && C >= 2) {
        if (tid == 0) {
            uint32_t localMode = prm.oscMode, pair = prm.trigPair;
            if (localMode == SGZ_OSC_MIDSIDE) { localMode = SGZ_OSC_MID; pair = prm.trigSeparate & ~1u; }     // :340-352
            const float *a, *b;
            if (localMode == SGZ_OSC_RIGHT) a = b = blk + size_t(pair + 1) * n;
            else if (localMode == SGZ_OSC_LEFT) a = b = blk + size_t(pair) * n;
            else if (localMode == SGZ_OSC_SEPARATE) a = b = blk + size_t(prm.trigSeparate) * n;
            else { a = blk + size_t(pair) * n; b = a + n; }
            double state = st->state;
            const double thr2 = st->threshold * st->threshold, hysteresis = st->hysteresis;
            int holding = st->isPeakHold;
            unsigned int qc = st->qCount;
            const unsigned int q0 = st->qHead;
            unsigned long long dropped = 0;
            for (uint32_t i = 0; i < n; ++i) {
                double sample = trigSample(localMode, a, b, i);
                sample *= sample;
                const double delta = sample - state;
                if (delta < 0) {
                    state *= 0.9999;
                    state = fmax(thr2, state);
                    if (holding) {
                        if (qc < kPeakCap) { prm.peaks[(q0 + qc) % kPeakCap] = playhead + (unsigned long long)i - 1ull; ++qc; }
                        else ++dropped;
                        holding = 0;
                    }
                } else {
                    if (delta > hysteresis * state) holding = 1;
                    state = sample;
                }
            }
            st->state = state; st->isPeakHold = holding; st->qCount = qc; st->droppedPeaks += dropped;
        }
        __syncthreads();
    }
    // ---- A: ZeroCrossingProcessor over the block (executeSamplingWindows, OscilloscopeDSP.inl:311-385)
    if (batchedA) {                                                        // (found for the whole batch above: this callback's triggers become visible)
        if (tid == 0) st->qCount += sFires[blockIndex];
        __syncthreads();
    } else if (prm.triggerMode == 4u && C >= 2) {
        uint32_t localMode = prm.oscMode, pair = prm.trigPair;
        if (localMode == SGZ_OSC_MIDSIDE) { localMode = SGZ_OSC_MID; pair = prm.trigSeparate & ~1u; }     // :340-352
        const float *a, *b;
        if (localMode == SGZ_OSC_RIGHT) a = b = blk + size_t(pair + 1) * n;
        else if (localMode == SGZ_OSC_LEFT) a = b = blk + size_t(pair) * n;
        else if (localMode == SGZ_OSC_SEPARATE) a = b = blk + size_t(prm.trigSeparate) * n;
        else { a = blk + size_t(pair) * n; b = a + n; }
        const double threshold = st->threshold, prevState = st->state;
        const int armedIn = st->isPeakHold;
        const unsigned long long originIn = st->crossOrigin;
        const uint32_t seg = (n + T - 1) / T;
        const uint32_t i0 = min(n, uint32_t(tid) * seg), i1 = min(n, i0 + seg);
        // arm_i = (s_i > 0 && s_{i-1} < 0); fire_i <=> s_i > threshold && lastArm(i) > lastThr(i-1)  (virtual indices: an arm
        // inherited from the previous block sits at -1, "no arm" at -3, "no threshold crossing yet" at -2)
        int segArm = INT_MIN, segThr = INT_MIN;                           // (sample indices of this block, < 2^31; "nothing yet" below every sentinel)
        for (uint32_t i = i0; i < i1; ++i) {
            const double s = trigSample(localMode, a, b, i);
            const double prev = i ? trigSample(localMode, a, b, i - 1) : prevState;
            if (s > 0 && prev < 0) segArm = int(i);
            if (s > threshold) segThr = int(i);
        }
        int totArm, totThr, inArm, inThr;
        blockExclusiveMax2(segArm, armedIn ? -1 : -3, segThr, -2, sScan2, inArm, inThr, totArm, totThr);
        int la = inArm, lt = inThr;
        unsigned int fires = 0;
        for (uint32_t i = i0; i < i1; ++i) {
            const double s = trigSample(localMode, a, b, i);
            const double prev = i ? trigSample(localMode, a, b, i - 1) : prevState;
            if (s > 0 && prev < 0) la = int(i);
            if (s > threshold) { if (la > lt) ++fires; lt = int(i); }
        }
        unsigned int totalFires;
        unsigned int pos = blockExclusiveSum(fires, sSum, &totalFires);
        const unsigned int q0 = st->qHead, qc0 = st->qCount;
        la = inArm; lt = inThr;
        for (uint32_t i = i0; i < i1; ++i) {
            const double s = trigSample(localMode, a, b, i);
            const double prev = i ? trigSample(localMode, a, b, i - 1) : prevState;
            if (s > 0 && prev < 0) la = int(i);
            if (s > threshold) {
                if (la > lt) {
                    if (qc0 + pos < kPeakCap) prm.peaks[(q0 + qc0 + pos) % kPeakCap] = (la == -1) ? originIn : playhead + (unsigned long long)la;
                    ++pos;
                }
                lt = int(i);
            }
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned int room = kPeakCap - qc0;
            st->qCount = qc0 + (totalFires < room ? totalFires : room);
            if (totalFires > room) st->droppedPeaks += totalFires - room;
            st->isPeakHold = totArm > totThr ? 1 : 0;
            if (totArm >= 0) st->crossOrigin = playhead + (unsigned long long)totArm;
            st->state = n ? trigSample(localMode, a, b, n - 1) : prevState;
        }
        __syncthreads();
    }

    ICLK(1);
    // ---- B: processMutating's automaton (thread 0) -> swap list.  The head of the trigger queue is fetched by 64 lanes at once and the
    // swap list starts in LDS: the automaton then walks without memory round trips (a block holds a handful of triggers)
    if (hold && tid < int(kStage) && uint32_t(tid) < st->qCount) sPeaks[tid] = prm.peaks[(st->qHead + uint32_t(tid)) % kPeakCap];
    const unsigned int qHeadIn = st->qHead;
    __syncthreads();
    auto peakAt = [&](unsigned int q) {
        const unsigned int k = (q + kPeakCap - qHeadIn) % kPeakCap;
        return k < kStage ? sPeaks[k] : prm.peaks[q];
    };
    if (tid < 64) {                                                  // wave 0, every lane the same scalar walk (lane 0 stores); the rounds below use all of them
        const bool lane0 = tid == 0;
        unsigned int numSwaps = 0, lastStart = 0, lastLen = n;
        const unsigned long long written0 = st->written;
        if (lane0) { sCursor0 = st->frontCursor; sWritten0 = written0; }
        if (hold) {
            unsigned long long bufferedSamples = st->bufferedSamples, frontOrigin = st->frontOrigin, steadyClock = st->steadyClock;
            unsigned long long oldPeak = st->oldPeak, currentPeak = st->currentPeak;
            unsigned int qHead = st->qHead, qCount = st->qCount;
            int isWorkingOnPeak = st->isWorkingOnPeak;
            unsigned int swapsDone = 0;
            unsigned long long numSamples = n, consumed = 0;
            if (frontOrigin + bufferedSamples < steadyClock) { frontOrigin = steadyClock; bufferedSamples = 0; }   // :81-85
            const double ceilingSize = ceil(st->windowSize);
            const double halfSize = ceilingSize / 2;
            const unsigned long long bufferedCap = (unsigned long long)(ceilingSize + 1);                      // :101
            // (for an integer d >= 0:  double(d) < halfSize  <=>  d < ceil(halfSize))
            const unsigned long long halfCeil = (unsigned long long)ceil(halfSize);
            auto processIntoBackBuffer = [&](unsigned long long samples) {                                     // :90-105
                lastStart = (unsigned int)consumed; lastLen = (unsigned int)samples;
                consumed += samples;
                numSamples -= samples;
                const unsigned long long oldSamples = bufferedSamples;
                steadyClock += samples;
                bufferedSamples += samples;
                bufferedSamples = minU64(bufferedSamples, bufferedCap);
                frontOrigin += (oldSamples + samples) - bufferedSamples;
            };
            if (ceilingSize == 0 && qCount) qCount = 0;                                                        // :107-110
            // The walk below is the reference's, branch for branch.  In front of it, the same walk for the case a running stream is
            // in at almost every trigger -- the next trigger lies inside the current window (less than half a window after the last
            // one): what the general body computes then is  missing = peak - oldPeak, neededPreSamples = 0, take what is missing
            // from the callback, swap min(buffered, missing + 1) samples  (every
            // double in :147-190 is an integer plus, at most, the half of an odd window: the casts truncate it away on both sides of
            // the one subtraction that matters).  Some forty triggers per callback at cfg3: 8.9 us of a 16 us callback in the general body, 6.3 us here (a lone lane issues an
            // instruction every ~8 cycles; the same in 32-bit arithmetic relative to the callback's start, or with the queue's head read
            // one trigger ahead, measured no faster).  Anything else -- a trigger outside the window -- leaves the state exactly as
            // the general body expects it at that point and falls through to it.
            while (numSamples != 0) {
                // ---- a ROUND: up to 64 queued triggers at once, one per lane.  For a trigger that is fetched with the running clock
                // already past it (no samples swallowed at the fetch, :128), inside the window and in order, the walk does
                //     taken = max(missing - buffered, 0) samples from the callback;  swap min(max(buffered, missing), missing + 1);
                //     buffered <- max(buffered - (missing + 1), 0)
                // -- maps that compose by addition: buffered in front of trigger l is max(buffered_0 - sum_{i<l}(missing_i + 1), 0),
                // so every trigger's numbers are prefix sums.  The round takes the longest run of such triggers that the callback's
                // samples cover (the first one fetched at or after the running clock, out of order, outside the window, or short of
                // samples ends it) and leaves the walk's state behind it; the scalar iteration below handles whatever comes next.
                if (!isWorkingOnPeak && qCount >= 4u && halfCeil <= (1ull << 24) && bufferedSamples <= bufferedCap && bufferedSamples < (1ull << 30) &&
                    numSamples < (1ull << 30) && steadyClock < (1ull << 51)) {
                    const unsigned long long base = steadyClock;
                    auto fits = [&](unsigned long long v) { const long long r = (long long)(v - base); return r > -(1ll << 30) && r < (1ll << 30); };
                    if (fits(oldPeak)) {
                        const unsigned int lane = (unsigned int)tid, cnt = qCount < 64u ? qCount : 64u;
                        const unsigned int buf0 = (unsigned int)bufferedSamples, ns0 = (unsigned int)numSamples, hc = (unsigned int)halfCeil;
                        auto scan = [&](unsigned int v) {                         // inclusive prefix sum over the wave
#pragma unroll
                            for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(v, o); if (lane >= (unsigned int)o) v += u; }
                            return v;
                        };
                        const unsigned long long pk = lane < cnt ? peakAt((qHead + lane) % kPeakCap) : 0ull;
                        bool ok = lane < cnt && fits(pk);
                        const int p = ok ? int((long long)(pk - base)) : 0;
                        int prev = __shfl_up(p, 1);
                        if (lane == 0) prev = int((long long)(oldPeak - base));
                        ok = ok && p >= prev;
                        const unsigned int d = ok ? (unsigned int)(p - prev) : 0u;
                        ok = ok && d < hc;
                        const unsigned int a = ok ? d + 1u : 0u;
                        const unsigned int S = scan(a), Sprev = S - a;
                        const unsigned int bufL = buf0 > Sprev ? buf0 - Sprev : 0u;
                        const unsigned int tp = ok && d > bufL ? d - bufL : 0u;
                        const unsigned int Cc = scan(tp), Cprev = Cc - tp;
                        const bool swallow = ok && p >= 0 && (unsigned int)p >= Cprev;       // fetched at or after the running clock (base + Cprev)
                        const bool stop = !ok || swallow || Cc > ns0 || Cprev >= ns0;
                        const unsigned long long stops = __ballot(stop);
                        const unsigned int m = stops ? (unsigned int)__builtin_ctzll(stops) : 64u;   // triggers 0 .. m-1 complete
                        if (m > 0) {
                            const unsigned int bufTake = bufL > d ? bufL : d, capped = bufTake < d + 1u ? bufTake : d + 1u;
                            const unsigned int capSum = scan(lane < m ? capped : 0u);
                            if (lane < m && numSwaps + lane < kMaxSwaps) {
                                const Swap sw{written0 + consumed + Cc - bufTake, capped};
                                if (numSwaps + lane < kStage) sSwaps[numSwaps + lane] = sw; else prm.swapList[numSwaps + lane] = sw;
                            }
                            const unsigned long long took = __ballot(lane < m && tp > 0u);
                            if (took) {
                                const int hl = 63 - __builtin_clzll(took);
                                lastStart = (unsigned int)consumed + __shfl(Cprev, hl);
                                lastLen = __shfl(tp, hl);
                            }
                            const unsigned int Sm = __shfl(S, int(m) - 1), Cm = __shfl(Cc, int(m) - 1), capM = __shfl(capSum, int(m) - 1);
                            const int pm = __shfl(p, int(m) - 1);
                            bufferedSamples = buf0 > Sm ? buf0 - Sm : 0u;
                            numSamples -= Cm; consumed += Cm; steadyClock += Cm;
                            frontOrigin += capM;
                            oldPeak = currentPeak = base + (unsigned long long)(long long)pm;
                            qHead = (qHead + m) % kPeakCap; qCount -= m;
                            swapsDone += m;
                            numSwaps = numSwaps + m < kMaxSwaps ? numSwaps + m : kMaxSwaps;
                            continue;
                        }
                    }
                }
                bool fastDone = false;
                if (qCount) do {
                    if (!isWorkingOnPeak) {                                          // :120-141
                        const unsigned long long nextPeak = peakAt(qHead);
                        isWorkingOnPeak = 1;
                        if (nextPeak >= steadyClock) {
                            const unsigned long long deltaToPeak = nextPeak - steadyClock;
                            const unsigned long long toProcess = minU64(numSamples, (unsigned long long)(double(deltaToPeak) + halfSize));
                            processIntoBackBuffer(toProcess);
                        }
                        currentPeak = nextPeak;
                    }
                    if (currentPeak < oldPeak) break;                                // (the general body: the difference wraps, "outside the window")
                    const unsigned long long d = currentPeak - oldPeak;
                    if (d >= halfCeil || oldPeak >= (1ull << 51)) break;
                    // inside the window: windowEnd = oldPeak + (u64)halfSize, peakWindowEnd = currentPeak + (u64)halfSize,
                    // missingBufferSamples = d, neededPreSamples = 0
                    if (bufferedSamples < d) {                                       // :176-182: the samples still missing come out of this callback
                        const unsigned long long numRemaining = d - bufferedSamples;
                        const unsigned long long toProcess = minU64(numSamples, numRemaining);
                        if (toProcess > 0) processIntoBackBuffer(toProcess);
                        if (numRemaining != toProcess) { fastDone = true; break; }   // not ready: the callback is used up, the trigger stays open
                    }
                    const unsigned long long cappedSize = minU64(bufferedSamples, d + 1ull);
                    if (numSwaps < kMaxSwaps) {
                        const Swap sw{(written0 + consumed) - bufferedSamples, (unsigned int)cappedSize};
                        if (lane0) { if (numSwaps < kStage) sSwaps[numSwaps] = sw; else prm.swapList[numSwaps] = sw; }     // (two stores: a selected reference is a flat one)
                        ++numSwaps;
                    }
                    bufferedSamples -= cappedSize;
                    frontOrigin += cappedSize;
                    oldPeak = currentPeak;
                    isWorkingOnPeak = 0;
                    qHead = (qHead + 1) % kPeakCap; qCount--;
                    ++swapsDone;
                    fastDone = true;
                } while (false);
                if (fastDone) continue;
                if (!qCount) { processIntoBackBuffer(numSamples); break; }
                else if (!isWorkingOnPeak) {
                    isWorkingOnPeak = 1;
                    const unsigned long long nextPeak = peakAt(qHead);
                    if (nextPeak >= steadyClock) {
                        const unsigned long long deltaToPeak = nextPeak - steadyClock;
                        const unsigned long long toProcess = minU64(numSamples, (unsigned long long)(double(deltaToPeak) + halfSize));
                        processIntoBackBuffer(toProcess);
                        currentPeak = nextPeak;
                    } else currentPeak = nextPeak;
                }
                unsigned long long windowEnd;
                bool isPeakOutsideOfWindow = false, readyForBufferSwap = false;
                if (double(currentPeak - oldPeak) < halfSize) windowEnd = (unsigned long long)(double(oldPeak) + halfSize);   // :147-150
                else { isPeakOutsideOfWindow = true; windowEnd = frontOrigin + bufferedSamples; }
                const unsigned long long peakWindowEnd =
                    (unsigned long long)(double((isPeakOutsideOfWindow ? 1ull : 0ull) + currentPeak) + halfSize);       // :158
                unsigned long long toProcess = 0;
                const unsigned long long missingBufferSamples = peakWindowEnd - minU64(peakWindowEnd, windowEnd);
                const unsigned long long neededPreSamples =
                    minU64((unsigned long long)halfSize, (unsigned long long)(fmax(double(currentPeak - oldPeak), halfSize) - halfSize));
                if (isPeakOutsideOfWindow) {
                    toProcess = minU64(numSamples, missingBufferSamples);
                    if (toProcess > 0) processIntoBackBuffer(toProcess);
                    readyForBufferSwap = missingBufferSamples == toProcess;
                } else if (bufferedSamples >= missingBufferSamples) readyForBufferSwap = true;
                else {
                    const unsigned long long numRemaining = missingBufferSamples - bufferedSamples;
                    toProcess = minU64(numSamples, numRemaining);
                    if (toProcess > 0) processIntoBackBuffer(toProcess);
                    readyForBufferSwap = numRemaining == toProcess;
                }
                if (readyForBufferSwap) {
                    const double amount = (isPeakOutsideOfWindow ? halfSize : double(missingBufferSamples)) + double(neededPreSamples);
                    const unsigned long long cappedSize = minU64(bufferedSamples, (unsigned long long)ceil(amount + 1));
                    // swapBuffers(cappedSize, -bufferedSamples): source = the oldest buffered sample onwards
                    if (numSwaps < kMaxSwaps) {
                        const Swap sw{(written0 + consumed) - bufferedSamples, (unsigned int)cappedSize};
                        if (lane0) { if (numSwaps < kStage) sSwaps[numSwaps] = sw; else prm.swapList[numSwaps] = sw; }
                        ++numSwaps;
                    }
                    bufferedSamples -= minU64(bufferedSamples, cappedSize);
                    frontOrigin += cappedSize;
                    oldPeak = currentPeak;
                    isWorkingOnPeak = 0;
                    if (qCount) { qHead = (qHead + 1) % kPeakCap; qCount--; }
                    ++swapsDone;
                }
            }
            if (lane0) {
                st->swaps += swapsDone;                            // (once: a read-modify-write of global memory inside the walk is a round trip per trigger)
                st->bufferedSamples = bufferedSamples; st->frontOrigin = frontOrigin; st->steadyClock = steadyClock;
                st->oldPeak = oldPeak; st->currentPeak = currentPeak; st->qHead = qHead; st->qCount = qCount;
                st->isWorkingOnPeak = isWorkingOnPeak;
            }
        }
        if (lane0) { sNumSwaps = numSwaps; sLastStart = lastStart; sLastLen = lastLen; }
    }
    __syncthreads();

    ICLK(2);
    // ---- F: per-sample colours of the block (audioProcessing :445-517, :588-647).  Every sample passes through audioProcessing exactly
    // once and in order, whatever the split into calls, so the filters run over the block as a whole.
    if (prm.colours) {
        // The three filter stages hand their per-sample outputs on through LDS, a tile of the callback at a time (samples in, band
        // signals, smoothed energies: 11 floats per channel and sample; a stereo tile is 416 samples).  Through the HBM scratch every
        // sixteen samples of every stage waited for a memory round trip of its own: 64 us per stage, 193 us of a 206 us callback.
        constexpr uint32_t kFFloats = 9216;
        __shared__ float sF[kFFloats];
        const uint32_t MB = col.maxBlock;
        ColourDev *cs = col.st;
        uint32_t tile = kFFloats / (11u * C);
        tile = tile >= 16u ? (tile & ~15u) : (tile ? tile : 1u);
        if (tile > 512u) tile = 512u;
        float *sX = sF, *sB = sX + size_t(C) * tile, *sS = sB + size_t(4 * C) * tile;       // [C][tile], [4 C][tile], [6 C][tile]
        // the lanes' filter states stay in registers from tile to tile
        const bool r12 = uint32_t(tid) < 2 * C, r3 = uint32_t(tid) < 6 * C;
        const uint32_t fc = uint32_t(tid) >> 1, fhp = uint32_t(tid) & 1u;
        float k1[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, k2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f, d0 = 0.f, d1 = 0.f, y = 0.f;
        float *stp = nullptr;
        const uint32_t pair3 = uint32_t(tid) / 12u, r = uint32_t(tid) % 12u, sig3 = r / 3u, band3 = r % 3u;
        if (r12) {
            const float *ka = fhp ? col.hp1 : col.lp1, *kb = fhp ? col.hp2 : col.lp2;
            for (int j = 0; j < 5; ++j) { k1[j] = ka[j]; k2[j] = kb[j]; }
            float (*z1)[2] = cs->z[fc] + (fhp ? 2 : 0), (*z2)[2] = cs->z[fc] + (fhp ? 6 : 4);
            a0 = z1[0][0]; a1 = z1[0][1]; b0 = z1[1][0]; b1 = z1[1][1];
            c0 = z2[0][0]; c1 = z2[0][1]; d0 = z2[1][0]; d1 = z2[1][1];
        }
        if (r3) {
            stp = sig3 == 0 ? &cs->smooth[2 * pair3][band3] : sig3 == 1 ? &cs->smooth[2 * pair3 + 1][band3]
                : sig3 == 2 ? &cs->aux[2 * pair3][band3] : &cs->aux[2 * pair3 + 1][band3];
            y = *stp;
        }
        const float pole = col.pole;
        for (uint32_t t0 = 0; t0 < n; t0 += tile) {
            const uint32_t m = min(tile, n - t0);
            for (uint32_t e = tid; e < C * m; e += T) { const uint32_t c = e / m, i = e - c * m; sX[size_t(c) * tile + i] = blk[size_t(c) * n + t0 + i]; }
            __syncthreads();
            if (r12) {                                                     // F1: low = LP4_f1(x), rest = HP4_f1(x)
                const float *x = sX + size_t(fc) * tile;
                float *out = sB + (size_t(fc) * 4 + (fhp ? 3 : 0)) * tile;
                walkSequential(m, [&](uint32_t i) { return x[i]; },
                               [&](uint32_t i, float v) { out[i] = biquadStep(k1, b0, b1, biquadStep(k1, a0, a1, v)); });
            }
            __syncthreads();
            if (r12) {                                                     // F2: mid = LP4_f2(rest), high = HP4_f2(rest)
                const float *x = sB + (size_t(fc) * 4 + 3) * tile;
                float *out = sB + (size_t(fc) * 4 + (fhp ? 2 : 1)) * tile;
                walkSequential(m, [&](uint32_t i) { return x[i]; },
                               [&](uint32_t i, float v) { out[i] = biquadStep(k2, d0, d1, biquadStep(k2, c0, c1, v)); });
            }
            __syncthreads();
            if (r3) {                                                      // F3: filterStates (:460-468) of left, right, mid, side
                const float *l = sB + (size_t(2 * pair3) * 4 + band3) * tile, *rr = sB + (size_t(2 * pair3 + 1) * 4 + band3) * tile;
                float *out = sS + size_t(tid) * tile;
                // (which signal a lane smooths is a lane constant: both inputs are read and the lane's one picked by selects -- as a
                // branch inside the batched loads the four cases ran one after the other: 90 us of a 150 us callback)
                const bool isL = sig3 == 0, isR = sig3 == 1, isMid = sig3 == 2;
                walkSequential(m, [&](uint32_t i) { return l[i]; },
                               [&](uint32_t i, float lv) {
                                   const float rv = rr[i];
                                   const float sum = lv + rv, dif = lv - rv;
                                   const float v = isL ? lv : (isR ? rv : (isMid ? sum : dif));
                                   const float input = v * v;
                                   y = input + pole * (y - input);
                                   out[i] = y;
                               });
            }
            __syncthreads();
            for (uint32_t e = tid; e < 2 * C * m; e += T) {                // F4: accumulateColour per (signal, sample)
                const uint32_t q = e / m, i = e - q * m, pair = q >> 2, sig = q & 3u;
                const float *sp = sS + size_t(pair * 12 + sig * 3) * tile + i;
                const float stv[3] = {sp[0], sp[tile], sp[2 * size_t(tile)]};
                const uint32_t keyCh = 2 * pair + (sig & 1u);              // left / mid: the left key, right / side: the right key
                const uint32_t plane = (sig < 2 ? 0u : C) + keyCh;         // cwLeft, cwRight -> colourData; cwMid, cwSide -> auxColourData
                col.block[size_t(plane) * MB + t0 + i] = accumulateColour(stv, col.band, col.keys[keyCh], col.blend);
            }
            __syncthreads();
        }
        if (r12) {
            float (*z1)[2] = cs->z[fc] + (fhp ? 2 : 0), (*z2)[2] = cs->z[fc] + (fhp ? 6 : 4);
            z1[0][0] = a0; z1[0][1] = a1; z1[1][0] = b0; z1[1][1] = b1;
            z2[0][0] = c0; z2[0][1] = c1; z2[1][0] = d0; z2[1][1] = d1;
        }
        if (r3) *stp = y;
        __syncthreads();
    }

    // ---- C: swaps (ZeroCrossing) or the block itself (None / Spectral: audioProcessing straight into the front buffer), in order
    const uint32_t size = prm.size;
    uint32_t cursor = sCursor0;
    ICLK(3);
    const unsigned long long written0 = sWritten0;
    // `later`: samples that swaps after this one (same block) will append.  The ring holds `size` samples, so whatever this swap writes
    // survives only where fewer than `size` samples follow: a noisy signal fires a dozen triggers per block, each swap copies a
    // window's worth, and only the last one or two are still there when the block is done -- the others just move the cursor.
    auto appendFront = [&](unsigned long long src, uint32_t len, unsigned long long later) {
        // only the last `size` samples of a longer run survive in the ring
        uint32_t skip = len > size ? len - size : 0;
        const uint32_t cur0 = uint32_t((cursor + skip) % size);
        // of the remaining m samples the first `dead` are overwritten by the later swaps
        const uint32_t m0 = len - skip;
        const unsigned long long room = later >= size ? 0ull : size - later;          // newest samples of this swap that stay visible
        const uint32_t dead = m0 > room ? uint32_t(m0 - room) : 0u;
        const uint32_t m = m0 - dead;
        for (uint32_t c = 0; c < C; ++c)
            for (uint32_t i = tid; i < m; i += T) {
                const unsigned long long abs = src + skip + dead + i;
                const float v = abs >= written0 ? blk[size_t(c) * n + uint32_t(abs - written0)]
                                                : prm.back[size_t(c) * prm.backCap + uint32_t(abs & (prm.backCap - 1))];
                uint32_t d = (cur0 + dead + i) % size;
                prm.front[size_t(c) * size + d] = v;
            }
        if (prm.colours)
            for (uint32_t c = 0; c < 2 * C; ++c)
                for (uint32_t i = tid; i < m; i += T) {
                    const unsigned long long abs = src + skip + dead + i;
                    const uint32_t v = abs >= written0 ? col.block[size_t(c) * col.maxBlock + uint32_t(abs - written0)]
                                                       : col.back[size_t(c) * prm.backCap + uint32_t(abs & (prm.backCap - 1))];
                    uint32_t d = (cur0 + dead + i) % size;
                    col.front[size_t(c) * size + d] = v;
                }
        cursor = uint32_t((cursor + len) % size);
        __syncthreads();
    };
    if (hold && sNumSwaps <= kFlat) {
        // The usual case -- a handful of swaps, all in LDS -- as ONE pass: no two swaps of a block write the same ring slot (what a later
        // swap would overwrite is `dead` and never written), and their sources are read-only here, so they need no order among
        // themselves.  Lane k of wave 0 works out swap k's surviving range from prefix sums of the lengths; then every thread
        // copies its share of the concatenation.  (One swap after the other with a barrier each cost a memory round trip per swap:
        // 8.9 us of a 25 us block at a dozen swaps, tools/ingest_clocks.py; now 1.9 us.)
        const uint32_t ns = sNumSwaps;
        if (ns > kStage) {
            // more swaps than a wave has lanes (a 10 kHz tone at 48 kHz crosses zero 107 times per 512-sample callback): the same
            // tables from workgroup-wide prefix sums, one swap per thread (the swaps beyond the staged ones are read back from HBM).
            // One after the other they cost 0.65 us each: 74 us per callback at 107, 143 us at 213.
            const uint32_t k = uint32_t(tid);
            const bool live = k < ns;
            Swap sw{0ull, 0u, 0u};
            if (live) sw = k < kStage ? sSwaps[k] : prm.swapList[k];
            const uint32_t len = sw.len;
            unsigned long long total;
            const unsigned long long inc = blockInclusiveSum<unsigned long long>(len, sScan64, &total);
            const unsigned long long before = inc - len, later = total - inc;
            const uint32_t skip = len > size ? len - size : 0u;
            const uint32_t m0 = len - skip;
            const unsigned long long room = later >= size ? 0ull : size - later;
            const uint32_t dead = m0 > room ? uint32_t(m0 - room) : 0u;
            const uint32_t m = m0 - dead;
            uint32_t totalM;
            const uint32_t incM = blockInclusiveSum<uint32_t>(m, sSum, &totalM);
            if (live) {
                sFlatSrc[k] = sw.src + skip + dead;
                sFlatDst[k] = uint32_t((sCursor0 + before + skip + dead) % size);
                sFlatEnd[k] = incM;
            }
            if (tid == 0) { sFlatTotal = totalM; sCursorEnd = uint32_t((sCursor0 + total) % size); }
        } else if (tid < 64) {
            const uint32_t k = uint32_t(tid);
            const bool live = k < ns;
            const uint32_t len = live ? sSwaps[live ? k : 0].len : 0u;
            const unsigned long long src0 = live ? sSwaps[k].src : 0ull;
            unsigned long long inc = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = __shfl_up(inc, o); if (tid >= o) inc += u; }
            const unsigned long long total = __shfl(inc, 63), before = inc - len, later = total - inc;
            const uint32_t skip = len > size ? len - size : 0u;
            const uint32_t m0 = len - skip;
            const unsigned long long room = later >= size ? 0ull : size - later;
            const uint32_t dead = m0 > room ? uint32_t(m0 - room) : 0u;
            const uint32_t m = m0 - dead;
            uint32_t incM = m;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incM, o); if (tid >= o) incM += u; }
            sFlatSrc[k] = src0 + skip + dead;
            sFlatDst[k] = uint32_t((sCursor0 + before + skip + dead) % size);
            sFlatEnd[k] = incM;
            if (tid == 63) { sFlatTotal = incM; sCursorEnd = uint32_t((sCursor0 + total) % size); }
        }
        __syncthreads();
        const uint32_t M = sFlatTotal;
        auto locate = [&](uint32_t e, unsigned long long &abs, uint32_t &d) {
            uint32_t lo = 0, hi = ns;                                    // smallest k with sFlatEnd[k] > e
            while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (sFlatEnd[mid - 1] > e) hi = mid; else lo = mid; }
            const uint32_t i = e - (lo ? sFlatEnd[lo - 1] : 0u);
            abs = sFlatSrc[lo] + i;
            d = (sFlatDst[lo] + i) % size;
        };
        for (uint32_t e = tid; e < M; e += T) {
            unsigned long long abs; uint32_t d;
            locate(e, abs, d);
            const bool fromBlock = abs >= written0;
            const uint32_t bi = uint32_t(abs - written0), ri = uint32_t(abs & (prm.backCap - 1));
            for (uint32_t c = 0; c < C; ++c)
                prm.front[size_t(c) * size + d] = fromBlock ? blk[size_t(c) * n + bi] : prm.back[size_t(c) * prm.backCap + ri];
            if (prm.colours)
                for (uint32_t c = 0; c < 2 * C; ++c)
                    col.front[size_t(c) * size + d] = fromBlock ? col.block[size_t(c) * col.maxBlock + bi] : col.back[size_t(c) * prm.backCap + ri];
        }
        cursor = sCursorEnd;
        __syncthreads();
    } else if (hold) {
        const uint32_t ns = sNumSwaps;
        // suffix sums of the swap lengths: thread-private walk from the back (the list is short)
        unsigned long long later = 0;
        auto swapAt = [&](uint32_t k) { return k < kStage ? sSwaps[k] : prm.swapList[k]; };
        for (uint32_t k = 0; k < ns; ++k) later += swapAt(k).len;
        for (uint32_t k = 0; k < ns; ++k) {
            const Swap sw = swapAt(k);
            later -= sw.len;
            appendFront(sw.src, sw.len, later);
        }
    } else appendFront(written0, n, 0ull);

    ICLK(4);
    // ---- D: the block goes into the back rings (ZeroCrossing / EnvelopeHold only; absolute index mod backCap)
    if (hold) {
        const uint32_t keep = n > prm.backCap ? prm.backCap : n, first = n - keep;
        for (uint32_t e = tid; e < keep * C; e += T) {
            const uint32_t c = e / keep, i = first + (e - c * keep);
            prm.back[size_t(c) * prm.backCap + uint32_t((written0 + i) & (prm.backCap - 1))] = blk[size_t(c) * n + i];
        }
        if (prm.colours)
            for (uint32_t e = tid; e < keep * 2 * C; e += T) {
                const uint32_t c = e / keep, i = first + (e - c * keep);
                col.back[size_t(c) * prm.backCap + uint32_t((written0 + i) & (prm.backCap - 1))] = col.block[size_t(c) * col.maxBlock + i];
            }
    }

    ICLK(5);
    // ---- E: RMS envelope (audioProcessing, OscilloscopeDSP.inl:520-585, :676-693); wave 0, lane c = recurrence c
    // (the reference runs the recurrence in PeakDecay mode too and throws the result away, :506-585 against :676: only RMS stores it)
    if (prm.envMode == 1u && tid < 64) {
        const bool active = tid < int(C);
        const float k = prm.envelopeCoeff;
        const uint32_t mode = prm.oscMode;
        const uint32_t lastStart = sLastStart, lastLen = sLastLen;
        // channels 0 / 1 are stored back (:690-691) and so run through every audioProcessing call of the block; the others restart
        // from their stored envelope in every call, so only the last call's samples matter for them
        const uint32_t from = tid < 2 ? 0u : lastStart, to = lastStart + lastLen;
        float y = active ? st->envelope[tid] : 0.f;
        const float *b0 = blk, *b1 = blk + n, *bc = blk + size_t(active ? tid : 0) * n;
        bool own = false;
        if (mode == SGZ_OSC_SEPARATE) {
            own = true;
            if (active) walkSequential(to - from, [&](uint32_t i) { return bc[from + i]; }, [&](uint32_t, float v) { const float s = v * v; y = s + k * (y - s); });
        } else if (mode == SGZ_OSC_MIDSIDE) {
            if (tid < 2) {
                own = true;
                // the mix itself is exact in either order of evaluation: (l +- r) once, squared, halved
                walkSequential(to, [&](uint32_t i) { return tid == 0 ? b0[i] + b1[i] : b0[i] - b1[i]; },
                               [&](uint32_t, float v) { const float s = 0.5f * (v * v); y = s + k * (y - s); });
            }
        } else if (tid == 0) {
            own = true;
            const float *src = mode == SGZ_OSC_RIGHT ? b1 : b0;
            // (one walk per mode: a branch inside the load would keep the batch of loads from being issued together)
            auto rms = [&](uint32_t, float v) { const float s = v * v; y = s + k * (y - s); };
            if (mode == SGZ_OSC_MID) walkSequential(to, [&](uint32_t i) { return 0.5f * (b0[i] + b1[i]); }, rms);
            else if (mode == SGZ_OSC_SIDE) walkSequential(to, [&](uint32_t i) { return 0.5f * (b0[i] - b1[i]); }, rms);
            else walkSequential(to, [&](uint32_t i) { return src[i]; }, rms);
        }
        // filterEnv[c] of the last call: copies of channel 0 (mono modes) / channel 1 (MidSide) where the channel has no recurrence
        const float y0 = __shfl(y, 0), y1 = __shfl(y, 1);
        const float fe = own ? y : (mode == SGZ_OSC_MIDSIDE ? y1 : y0);
        if (prm.envMode == 1u) {                           // RMS: gain and stored envelopes
            float start = active ? __builtin_sqrtf(fe) : 0.f;
            for (int o = 32; o > 0; o >>= 1) start = fmaxf(start, __shfl_xor(start, o));
            if (tid == 0) {
                st->envelopeGain = 1.0 / double(start);
                st->envelope[0] = y0;
                st->envelope[1] = (mode == SGZ_OSC_SEPARATE || mode == SGZ_OSC_MIDSIDE) ? y1 : y0;
            }
        }
    }
    __syncthreads();
    ICLK(6);
    if (tid == 0) {
        st->frontCursor = cursor;
        if (hold) st->written = written0 + n;
        st->playhead = playhead + n;
    }
    __syncthreads();
    }   // next block of the batch
    for (uint32_t w = tid; w < sizeof(ScopeDev) / 4; w += T) reinterpret_cast<uint32_t *>(prm.st)[w] = reinterpret_cast<const uint32_t *>(&sState)[w];
    ICLK(7);
    if (prm.doPeak) {                                            // (uniform) the rings and the state this launch wrote, as every lane sees them
        __threadfence();
        __syncthreads();
        scopePeakBody(prm.peak);
    }
}

// ---- Oscilloscope::runPeakFilter on the front rings (memory order, last size mod lanes slots dropped), every channel mode.
// A 1024-thread workgroup: scopePeakKernel, or the tail of scopeIngestKernel.
__device__ void scopePeakBody(const PeakParams &prm)
{
    __shared__ float sL[kMaxCh][16], sR[16];
    const uint32_t C = prm.channels, size = prm.size;
    const uint32_t mode = C == 1 ? uint32_t(SGZ_OSC_LEFT) : prm.mode;
    const uint32_t len = prm.len ? prm.len : size;
    const uint32_t stop = len - (len & (prm.lanes - 1));
    // memory index of slot i of the reference's ring: the ring itself, or (Spectral) the newest len samples of the larger ring
    const uint32_t first = prm.len ? (prm.st->frontCursor + size - len) % size : 0u;
    auto at = [&](uint32_t i) { uint32_t p = first + i; return p >= size ? p - size : p; };
    const int tid = threadIdx.x, wave = tid >> 6, waves = blockDim.x >> 6;
    auto blockMax = [&](float v, float *slot) {
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        if ((tid & 63) == 0) slot[wave] = v;
    };
    const float *L = prm.front, *R = prm.front + (C > 1 ? size : 0);
    if (mode <= SGZ_OSC_SIDE) {
        float m = 0.f;
        for (uint32_t j = tid; j < stop; j += blockDim.x) {
            const uint32_t i = at(j);
            float v;
            if (mode == SGZ_OSC_LEFT) v = L[i];
            else if (mode == SGZ_OSC_RIGHT) v = R[i];
            else if (mode == SGZ_OSC_MID) v = (L[i] + R[i]) * 0.5f;
            else v = (L[i] - R[i]) * 0.5f;
            m = fmaxf(fabsf(v), m);
        }
        blockMax(m, sL[0]);
    } else if (mode == SGZ_OSC_SEPARATE) {
        for (uint32_t c = 0; c < C; ++c) {
            float m = 0.f;
            const float *x = prm.front + size_t(c) * size;
            for (uint32_t j = tid; j < stop; j += blockDim.x) m = fmaxf(fabsf(x[at(j)]), m);
            blockMax(m, sL[c]);
        }
    } else {
        float ml = 0.f, mr = 0.f;
        for (uint32_t j = tid; j < stop; j += blockDim.x) {
            const uint32_t i = at(j);
            const float a = L[i] + R[i], b = L[i] - R[i];
            ml = fmaxf(fabsf(a * 0.5f), ml);
            mr = fmaxf(fabsf(b * 0.5f), mr);
        }
        blockMax(ml, sL[0]);
        blockMax(mr, sR);
    }
    __syncthreads();
    if (tid != 0) return;
    ScopeDev *st = prm.st;
    auto fold = [&](const float *slot) { float r = 0.f; for (int w = 0; w < waves; ++w) r = fmaxf(r, slot[w]); return r; };
    const double coeff = prm.coeff;
    if (mode <= SGZ_OSC_SIDE) {
        const double highest = double(fold(sL[0]));
        st->envelope[0] = float(fmax(double(st->envelope[0]) * coeff, highest * highest));
        if (C > 1) st->envelope[1] = float(fmax(double(st->envelope[1]) * coeff, highest * highest));
        for (uint32_t c = 2; c < C; ++c) st->envelope[c] = st->envelope[1];
    } else if (mode == SGZ_OSC_SEPARATE) {
        float running = 0.f;                                // vLMax is not reset between channels (:827-838)
        for (uint32_t c = 0; c < C; ++c) {
            running = fmaxf(running, fold(sL[c]));
            st->envelope[c] = fmaxf(float(double(st->envelope[c]) * coeff), running * running);   // std::max<float>(...)
        }
    } else {
        const double hl = double(fold(sL[0])), hr = double(fold(sR));
        st->envelope[0] = float(fmax(double(st->envelope[0]) * coeff, hl * hl));
        st->envelope[1] = float(fmax(double(st->envelope[1]) * coeff, hr * hr));
        for (uint32_t c = 2; c < C; ++c) st->envelope[c] = st->envelope[1];
    }
    float start = __builtin_sqrtf(st->envelope[0]);
    for (uint32_t c = 0; c < C; ++c) start = fmaxf(start, __builtin_sqrtf(st->envelope[c]));
    st->autoGain = 1.0 / double(start);
}
__global__ void __launch_bounds__(1024) scopePeakKernel(const PeakParams prm) { scopePeakBody(prm); }


// ---- Spectral triggering: Oscilloscope::calculateFundamentalPeriod + calculateTriggeringOffset (OscilloscopeDSP.inl:62-308)
struct BinRec { unsigned long long index; double value, offset; };           // BinRecord, Oscilloscope.h
struct SpectralDev {
    BinRec median[8];                        // medianTriggerFilter[i].record (value-initialised)
    unsigned long long medianPos;
    BinRec record;                           // triggerState.record
    double fundamental, cycleSamples, sampleOffset, phase;
    unsigned long long ringSize;             // resizeAudioStorage's size (ChannelData.h:107-128): in = of the previous frame, out = of this one
};
struct SpectralParams {
    SpectralDev *st;
    const float *ringA, *ringB; uint32_t evalMode, cap;
    const uint32_t *d_cursor;
    double windowSize, sampleRate, threshold, hysteresis, phaseOffsetDeg, quarterSemitone;
    double customFrequency;                  // state.customTrigger ? state.customTriggerFrequency : 0
    const double2 *tw;                       // [4096] exp(-2 pi i k / 8192)
};

__device__ __forceinline__ double recOmega(const BinRec &r) { return double(r.index) + r.offset; }
__device__ __forceinline__ uint32_t brev13(uint32_t i) { return __brev(i) >> 19; }
__device__ __forceinline__ uint32_t logicalPhys(long rel, uint32_t cursor, uint32_t cap, uint32_t len)
{
    long q = rel % long(len);
    if (q < 0) q += long(len);
    return uint32_t((long(cursor) + long(cap - len) + q) % long(cap));
}
__device__ __forceinline__ double evalD(const float *a, const float *b, uint32_t mode, uint32_t idx)
{
    if (mode == 1u) return double(0.5f * (a[idx] + b[idx]));
    if (mode == 2u) return double(0.5f * (a[idx] - b[idx]));
    return double(a[idx]);
}

// libstdc++'s std::nth_element(v, v + 4, v + 8, by index) -- see oracle/scope_spectral.c for why the algorithm itself matters
__device__ inline bool recLess(const BinRec &a, const BinRec &b) { return a.index < b.index; }
__device__ inline void recSwap(BinRec &a, BinRec &b) { const BinRec t = a; a = b; b = t; }
__device__ void nthElementByIndex(BinRec *v, int n, int nth)
{
    int first = 0, last = n;
    int depth = 0;
    for (int k = n; k > 1; k >>= 1) ++depth;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            for (int i = first; i <= nth; ++i)
                for (int j = i + 1; j < last; ++j)
                    if (recLess(v[j], v[i])) recSwap(v[i], v[j]);
            return;
        }
        --depth;
        const int mid = first + (last - first) / 2;
        {   // __move_median_to_first(first, first + 1, mid, last - 1)
            const int r = first, a = first + 1, b = mid, c = last - 1;
            if (recLess(v[a], v[b])) {
                if (recLess(v[b], v[c])) recSwap(v[r], v[b]);
                else if (recLess(v[a], v[c])) recSwap(v[r], v[c]);
                else recSwap(v[r], v[a]);
            } else if (recLess(v[a], v[c])) recSwap(v[r], v[a]);
            else if (recLess(v[b], v[c])) recSwap(v[r], v[c]);
            else recSwap(v[r], v[b]);
        }
        int f = first + 1, l = last;                       // __unguarded_partition(first + 1, last, pivot = first)
        for (;;) {
            while (recLess(v[f], v[first])) ++f;
            --l;
            while (recLess(v[first], v[l])) --l;
            if (!(f < l)) break;
            recSwap(v[f], v[l]);
            ++f;
        }
        if (f <= nth) first = f;
        else last = f;
    }
    for (int i = first + 1; i < last; ++i) {               // __insertion_sort
        const BinRec val = v[i];
        if (recLess(val, v[first])) {
            for (int j = i; j > first; --j) v[j] = v[j - 1];
            v[first] = val;
        } else {
            int j = i;
            while (recLess(val, v[j - 1])) { v[j] = v[j - 1]; --j; }
            v[j] = val;
        }
    }
}

// One workgroup; dynamic LDS: re[8192], im[8192] doubles.  The transform is an in-place radix-2 DIF (bit-reversed output); natural bin
// k sits at position brev13(k).  Only bins 0 .. 4096 are read afterwards, i.e. position 1 and the even positions, so |X[i]| of bin i
// (1 <= i < 4096) is parked at re[brev13(i) + 1] -- the slot of bin i + 4096.
__global__ void __launch_bounds__(1024) scopeSpectralKernel(const SpectralParams prm)
{
    extern __shared__ double spectralLds[];
    double *re = spectralLds, *im = spectralLds + 8192;
    __shared__ double sRed[2][16];
    __shared__ double sRadians, sSampleDifference;
    __shared__ long sOffset2;
    const int tid = threadIdx.x;
    SpectralDev *st = prm.st;
    const uint32_t cursor = *prm.d_cursor, cap = prm.cap, len = uint32_t(st->ringSize);
    constexpr uint32_t N = 8192;

    // what calculateFundamentalPeriod leaves behind and what calculateTriggeringOffset derives from it first (:256-270)
    auto settle = [&](const BinRec &rec, double fundamental) {
        st->record = rec;
        st->fundamental = fundamental;
        const double cycleSamples = prm.sampleRate / fundamental;
        st->cycleSamples = cycleSamples;
        const double tau = 6.283185307179586476925286766559;
        const double radians = tau * recOmega(rec) / double(N);
        const double offsetReal = fmax(double(N), prm.windowSize + cycleSamples);
        const unsigned long long offset = (unsigned long long)ceil(offsetReal);
        sRadians = radians;
        sSampleDifference = double(offset) - (prm.windowSize + cycleSamples);
        sOffset2 = long(offset);
    };
    if (prm.customFrequency > 0) {
        // state.customTrigger (:71-81): the user names the frequency -- no transform, no median filter; the Goertzel phase below runs on it
        if (tid == 0) {
            BinRec rec; rec.index = 0; rec.value = 1; rec.offset = prm.customFrequency / prm.sampleRate * double(N);
            settle(rec, prm.customFrequency);
        }
    } else {
    {   // transformBuffer[i] = eval.evaluateSampleInc() from -max(ceil(effectiveWindowSize), LookaheadSize) (:92-99)
        const long offset = long(fmax(ceil(prm.windowSize), double(N)));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = uint32_t(tid) + 1024u * j;
            re[i] = evalD(prm.ringA, prm.ringB, prm.evalMode, logicalPhys(-offset + long(i), cursor, cap, len));
            im[i] = 0.0;
        }
    }
    __syncthreads();
    for (int s = 0; s < 13; ++s) {                         // DustFFT_fwdDa: forward DFT, fp64
        const uint32_t half = 4096u >> s;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t b = uint32_t(tid) + 1024u * j;
            const uint32_t pos = b & (half - 1), i0 = ((b >> (12 - s)) << (13 - s)) + pos, i1 = i0 + half;
            const double2 w = prm.tw[pos << s];
            const double ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
            const double dr = ar - br, di = ai - bi;
            re[i0] = ar + br; im[i0] = ai + bi;
            re[i1] = dr * w.x - di * w.y; im[i1] = dr * w.y + di * w.x;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = uint32_t(tid) + 1024u * j;
        if (i >= 1) { const uint32_t p = brev13(i); re[p + 1] = hypot(re[p], im[p]); }      // std::abs(complex<double>)
    }
    __syncthreads();

    if (tid < 64) {                                        // the candidate scan (:134-181), 64 bins at a time
        const int lane = tid;
        auto mag = [&](uint32_t i) { return re[brev13(i) + 1]; };
        auto quadDelta = [&](uint32_t w) -> double {       // :113-124
            const uint32_t p0 = brev13(w), p1 = brev13(w + 1), pm = brev13(w == 0 ? 1 : w - 1);
            const double dre = re[p0] * 2.0 - re[pm] - re[p1], dim = im[p0] * 2.0 - im[pm] - im[p1];
            if ((dre + dim) == 0) return 0.0;
            const double nre = re[pm] - re[p1], nim = im[pm] - im[p1];
            return (nre * dre + nim * dim) / (dre * dre + dim * dim);
        };
        const double invHysteresis = 1 - prm.hysteresis, quarterSemitone = prm.quarterSemitone;
        BinRec max;
        max.index = 1;
        max.value = fmax(prm.threshold * double(N) / 6.0, mag(1));
        max.offset = quadDelta(1);
        uint32_t start = 2;
        while (start < (N >> 1)) {
            const uint32_t i = start + uint32_t(lane);
            const bool cand = i < (N >> 1) && invHysteresis * mag(i) > max.value * 2;      // candidate must be vastly better
            const unsigned long long mask = __ballot(cand);
            if (!mask) { start += 64; continue; }
            const uint32_t k = start + uint32_t(__ffsll((long long)mask) - 1);
            BinRec current{k, mag(k), 0.0};
            if (recOmega(max) > 0) {
                current.offset = quadDelta(k);
                const double factor = recOmega(current) / recOmega(max);
                const double sensivity = current.value / max.value;
                if (invHysteresis * sensivity > 20) max = current;
                else if (fabs(1 - factor) < invHysteresis * quarterSemitone) max = current;
                else {
                    const double multipleDeviation = fabs(factor - floor(factor + 0.5));
                    if (invHysteresis * fabs(multipleDeviation) > quarterSemitone) max = current;
                }
            } else {
                max = current;
                max.offset = quadDelta(uint32_t(max.index));
            }
            start = k + 1;
        }
        if (lane == 0) {                                   // the median of 8 (:183-214)
            BinRec localMedian[8];
            for (int i = 0; i < 8; ++i) localMedian[i] = st->median[i];
            st->median[st->medianPos] = max;
            st->medianPos = (st->medianPos + 1) & 7ull;
            nthElementByIndex(localMedian, 8, 4);
            const BinRec oldMedianBin = localMedian[4];
            if (oldMedianBin.index != ~0ull && fabs(recOmega(max) - recOmega(oldMedianBin)) > 0.5) max = oldMedianBin;
            double fundamental = prm.sampleRate * recOmega(max) / double(N);
            settle(max, fmax(5.0, fundamental));
        }
    }
    }
    __syncthreads();
    {   // cpl::dsp::goertzel: z = s[N-1] - exp(-i w) s[N-2] = sum_n x[n] exp(i w (N - 1 - n)) (oracle/scope_spectral.c); evaluated as the
        // sum, all threads, fp64
        const double radians = sRadians;
        const long offset = sOffset2;
        double zr = 0, zi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t n = uint32_t(tid) + 1024u * j;
            const double x = evalD(prm.ringA, prm.ringB, prm.evalMode, logicalPhys(-offset + long(n), cursor, cap, len));
            double sn, cs;
            sincos(radians * double(N - 1 - n), &sn, &cs);
            zr += x * cs; zi += x * sn;
        }
        for (int o = 32; o > 0; o >>= 1) { zr += __shfl_xor(zr, o); zi += __shfl_xor(zi, o); }
        if ((tid & 63) == 0) { sRed[0][tid >> 6] = zr; sRed[1][tid >> 6] = zi; }
    }
    __syncthreads();
    if (tid == 0) {
        double zr = 0, zi = 0;
        for (int w = 0; w < 16; ++w) { zr += sRed[0][w]; zi += sRed[1][w]; }
        const double tau = 6.283185307179586476925286766559;
        const double rotation = -sSampleDifference * sRadians;              // :272-278
        const double cr = cos(rotation), ci = -sin(rotation);
        const double wr = zr * cr - zi * ci, wi = zr * ci + zi * cr;
        double phase = tau - atan2(wi, wr);
        phase += st->record.offset * tau;
        phase -= 1.5707963267948966192313216916398;
        phase += tau * prm.phaseOffsetDeg / 360;
        phase = fmod(phase, tau);
        while (phase < 0) phase += tau;
        st->phase = phase;
        const double cycles = phase / tau;
        st->sampleOffset = cycles * prm.sampleRate / st->fundamental - 1;
        // resizeAudioStorage(Spectral) for this frame (ChannelData.h:113-117)
        const unsigned long long need = (unsigned long long)(0.5 + st->cycleSamples + ceil(prm.windowSize));
        st->ringSize = need > N ? need : N;
    }
}

}  // namespace

struct sgz_scope {
    sgz_scope_config cfg{};
    std::atomic<bool> deferSubmit{false};      // sgz_scope_set_option(SGZ_RT_OPT_DEFER_SUBMIT); read by whoever holds the batch flag
    std::atomic<bool> parkPushes{false};                  // ... (SGZ_RT_OPT_PARK_PUSHES): every push waits in the host FIFO for the next reader / flush
    std::mutex mu;                    // configure (consumer thread) against push (producer: try_lock only, never waits)
    hipStream_t stream = nullptr;
    BatchRing batch;                           // staged blocks waiting for their (one) ingest launch (rt_common.hpp)
    uint32_t maxBlock = 0;
    Backlog backlog;                           // blocks waiting for a staging slot (rt_common.hpp)
    ScopeDev *d_state = nullptr;
    unsigned long long *d_peaks = nullptr;
    Swap *d_swaps = nullptr;
    float *d_front = nullptr, *d_back = nullptr;
    uint32_t size = 0, backCap = 0;
    uint32_t trigSeparate = 0, trigPair = 0;
    float envelopeCoeff = 0.f;
    // frequency colouring (colour_by_frequency)
    ColourParams col{};
    ColourParams *d_col = nullptr;               // device copy of `col`, what the ingest kernel reads
    // Spectral triggering
    SpectralDev *d_spectral = nullptr;
    double2 *d_tw = nullptr;
    std::atomic<long long> transport{0};                 // cs.transportPosition (OscilloscopeDSP.inl:706): TriggeringMode::Window places the window by it
    sgz_trigger_state trig{};                             // triggerState as of the last sgz_scope_analyse (consumer thread)
    // vertex output (consumer side)
    float *d_xyz = nullptr; uint32_t *d_rgba = nullptr; size_t vertexCap = 0;
    void *h_out = nullptr; size_t hOutBytes = 0;          // pinned
    uint64_t busy = 0;
};

static void scopeFree(sgz_scope *s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    s->batch.release();
    s->backlog.release();
    for (void *p : {(void *)s->d_state, (void *)s->d_peaks, (void *)s->d_swaps, (void *)s->d_front, (void *)s->d_back, (void *)s->d_xyz,
                    (void *)s->d_rgba, (void *)s->col.st, (void *)s->col.bands, (void *)s->col.sm, (void *)s->col.block, (void *)s->col.front,
                    (void *)s->col.back, (void *)s->d_spectral, (void *)s->d_tw, (void *)s->d_col})
        if (p) (void)hipFree(p);
    if (s->h_out) (void)hipHostFree(s->h_out);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

static sgz_status scopeValidate(const sgz_scope_config *c)
{
    if (!(c->sample_rate >= 1) || !std::isfinite(c->sample_rate)) return fail(SGZ_EINVAL, "sample_rate");
    if (!std::isfinite(c->window_size) || c->window_size < 0 || c->window_size > double(1u << 26)) return fail(SGZ_EINVAL, "window_size");
    if (c->num_channels < 2 || (c->num_channels & 1) || c->num_channels > kMaxCh)
        return fail(SGZ_EINVAL, "num_channels must be even, 2..64 (OscilloscopeDSP.inl:318)");
    if (c->trigger_mode > SGZ_TRIG_ZERO_CROSSING) return fail(SGZ_EINVAL, "trigger_mode");
    if (c->trigger_mode == SGZ_TRIG_ENVELOPE_HOLD && (!(c->trigger_hysteresis >= 0) || !std::isfinite(c->trigger_hysteresis)))
        return fail(SGZ_EINVAL, "trigger_hysteresis");
    if (c->trigger_mode == SGZ_TRIG_SPECTRAL) {
        if (!(c->trigger_hysteresis >= 0) || !(c->trigger_hysteresis <= 1)) return fail(SGZ_EINVAL, "trigger_hysteresis outside 0..1");
        if (!std::isfinite(c->trigger_phase_offset)) return fail(SGZ_EINVAL, "trigger_phase_offset");
        if (c->custom_trigger && (!(c->custom_trigger_frequency > 0) || !(c->custom_trigger_frequency < c->sample_rate * 0.5)))
            return fail(SGZ_EINVAL, "custom_trigger_frequency must lie in (0, sample_rate / 2)");
        if (c->sample_rate / 5.0 + c->window_size > double(1u << 26)) return fail(SGZ_EINVAL, "ring too long");
    }
    if (c->colour_by_frequency) {
        if (!(c->frequency_colouring_blend >= 0) || !(c->frequency_colouring_blend <= 1)) return fail(SGZ_EINVAL, "frequency_colouring_blend");
        if (!(c->colour_smoothing_ms >= 0) || !std::isfinite(c->colour_smoothing_ms)) return fail(SGZ_EINVAL, "colour_smoothing_ms");
        if (!(c->sample_rate > 2 * 3000.0)) return fail(SGZ_EINVAL, "frequency colouring needs a sample rate above 6 kHz (crossovers at 300 / 3000 Hz)");
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (!std::isfinite(c->band_colours[i][j])) return fail(SGZ_EINVAL, "band_colours");
    }
    if (c->channel_mode > SGZ_OSC_MIDSIDE || c->envelope_mode > SGZ_ENV_PEAK_DECAY) return fail(SGZ_EINVAL, "enum value");
    if (c->interpolation > SGZ_SUBSAMPLE_LANCZOS) return fail(SGZ_EINVAL, "interpolation");
    if (!std::isfinite(c->trigger_threshold) || !std::isfinite(c->trigger_channel) || !(c->trigger_channel >= 1)) return fail(SGZ_EINVAL, "trigger");
    if (!std::isfinite(c->envelope_window) || c->envelope_window < 0) return fail(SGZ_EINVAL, "envelope_window");
    if (c->max_block > (1u << 17)) return fail(SGZ_EINVAL, "max_block above 131072 samples");
    return SGZ_OK;
}

// (re)allocates everything for a configuration; the device state starts as make_unique<TriggeringProcessor>() leaves it (zeroes)
// followed by setSettings(mode, window, threshold, hysteresis) (Oscilloscope.cpp:310)
static sgz_status scopeSetup(sgz_scope *s, const sgz_scope_config *cfg, bool fresh)
{
    sgz_status st = scopeValidate(cfg);
    if (st != SGZ_OK) return st;
    if (!s->stream) SGZ_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    const uint32_t C = cfg->num_channels;
    // ChannelData::resizeAudioStorage (ChannelData.h:107-128).  Spectral: the largest ring the reference can ask for (5 Hz floor of the
    // fundamental); the reference's ring of the frame is cut out of it by the readers (sgz_trigger_state::ring_size)
    uint32_t size = uint32_t(std::ceil(cfg->window_size + 1));                       // :121
    if (cfg->trigger_mode == SGZ_TRIG_SPECTRAL)
        size = uint32_t(std::max<size_t>(size_t(0.5 + cfg->sample_rate / 5.0 + std::ceil(cfg->window_size)), 8192));
    uint32_t backCap = 1; while (backCap < size) backCap <<= 1;
    const uint32_t maxBlock = cfg->max_block ? cfg->max_block : 8192u;
    const bool colours = cfg->colour_by_frequency != 0;
    const bool realloc = fresh || C != s->cfg.num_channels || size != s->size || maxBlock != s->maxBlock ||
                         colours != (s->cfg.colour_by_frequency != 0);
    ScopeDev h{};
    if (!fresh) SGZ_HIP(hipMemcpy(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost));
    if (realloc) {
        for (void **p : {(void **)&s->d_front, (void **)&s->d_back}) if (*p) { (void)hipFree(*p); *p = nullptr; }
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_front), size_t(C) * size * sizeof(float)));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_back), size_t(C) * backCap * sizeof(float)));
        SGZ_HIP(hipMemset(s->d_front, 0, size_t(C) * size * sizeof(float)));
        SGZ_HIP(hipMemset(s->d_back, 0, size_t(C) * backCap * sizeof(float)));
        // a slot takes a whole batch: the blocks of one rendered frame and more (at least 8192 samples)
        if ((st = s->batch.init(C, std::max<uint32_t>(maxBlock, 8192u))) != SGZ_OK) return st;
        s->maxBlock = maxBlock;
        // one second of audio may wait for the GPU (at least 32 blocks)
        if (!s->backlog.init(backlogFloats(C, cfg->sample_rate, maxBlock))) return fail(SGZ_ENOMEM, "out of memory (push backlog)");
        if (!s->d_state) {
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_state), sizeof(ScopeDev)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_peaks), size_t(kPeakCap) * sizeof(unsigned long long)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_swaps), size_t(kMaxSwaps) * sizeof(Swap)));
        }
        h.frontCursor = 0; h.written = 0;
        for (void **p : {(void **)&s->col.st, (void **)&s->col.bands, (void **)&s->col.sm, (void **)&s->col.block, (void **)&s->col.front,
                         (void **)&s->col.back})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (colours) {
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->col.st), sizeof(ColourDev)));
            SGZ_HIP(hipMemset(s->col.st, 0, sizeof(ColourDev)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->col.bands), size_t(C) * 4 * maxBlock * sizeof(float)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->col.sm), size_t(C) * 6 * maxBlock * sizeof(float)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->col.block), size_t(C) * 2 * maxBlock * sizeof(uint32_t)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->col.front), size_t(C) * 2 * size * sizeof(uint32_t)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->col.back), size_t(C) * 2 * backCap * sizeof(uint32_t)));
            SGZ_HIP(hipMemset(s->col.front, 0, size_t(C) * 2 * size * sizeof(uint32_t)));
            SGZ_HIP(hipMemset(s->col.back, 0, size_t(C) * 2 * backCap * sizeof(uint32_t)));
        }
        s->col.maxBlock = maxBlock;
    }
    if (colours) {
        // tuneCrossOver(300, 3000, sampleRate), tuneColourSmoothing(ms, sampleRate) (ChannelData.h:163-171).  cpl's designs are not in the
        // reference tree: the 3-band Linkwitz-Riley tree and the one-pole design are the published ones (oracle/scope_spectral.c)
        auto butterworth = [](double fnorm, bool highpass, float *c) {
            const double w0 = 6.283185307179586476925286766559 * fnorm;
            const double cw = std::cos(w0), alpha = std::sin(w0) / (2.0 * 0.70710678118654752440);
            const double a0 = 1 + alpha;
            double b0, b1;
            if (highpass) { b0 = (1 + cw) / 2; b1 = -(1 + cw); }
            else { b0 = (1 - cw) / 2; b1 = 1 - cw; }
            c[0] = float(b0 / a0); c[1] = float(b1 / a0); c[2] = float(b0 / a0);
            c[3] = float(-2 * cw / a0); c[4] = float((1 - alpha) / a0);
        };
        const float f1 = float(300.0 / cfg->sample_rate), f2 = float(3000.0 / cfg->sample_rate);
        butterworth(double(f1), false, s->col.lp1); butterworth(double(f1), true, s->col.hp1);
        butterworth(double(f2), false, s->col.lp2); butterworth(double(f2), true, s->col.hp2);
        s->col.pole = float(std::exp(-1.0 / (cfg->colour_smoothing_ms / 1000.0 * cfg->sample_rate)));
        s->col.blend = 1 - cfg->frequency_colouring_blend;                               // OscilloscopeDSP.inl:515
        std::memcpy(s->col.band, cfg->band_colours, sizeof(s->col.band));
        for (uint32_t c = 0; c < C; ++c) std::memcpy(&s->col.keys[c], cfg->colours[c], 4);
    }
    if (cfg->trigger_mode == SGZ_TRIG_SPECTRAL) {
        if (!s->d_tw) {
            std::vector<double2> tw(4096);
            for (int k = 0; k < 4096; ++k) {
                const double a = -6.283185307179586476925286766559 * double(k) / 8192.0;
                tw[k] = make_double2(std::cos(a), std::sin(a));
            }
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_tw), sizeof(double2) * 4096));
            SGZ_HIP(hipMemcpy(s->d_tw, tw.data(), sizeof(double2) * 4096, hipMemcpyHostToDevice));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_spectral), sizeof(SpectralDev)));
            SGZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scopeSpectralKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        2 * 8192 * int(sizeof(double))));
        }
        if (realloc || s->cfg.trigger_mode != SGZ_TRIG_SPECTRAL) {
            SpectralDev sd{};                                                             // Oscilloscope(): medianPos(), value-initialised filter
            sd.ringSize = std::max<size_t>(size_t(0.5 + 0.0 + std::ceil(cfg->window_size)), 8192);
            SGZ_HIP(hipMemcpy(s->d_spectral, &sd, sizeof(sd), hipMemcpyHostToDevice));
            s->trig = sgz_trigger_state{};
            s->trig.ring_size = sd.ringSize;
        }
    } else {
        s->trig = sgz_trigger_state{};
        s->trig.ring_size = size;
        if (cfg->trigger_mode == SGZ_TRIG_ZERO_CROSSING || cfg->trigger_mode == SGZ_TRIG_ENVELOPE_HOLD)   // calculateTriggeringOffset :233-240
            s->trig.sample_offset = (cfg->window_size * 0.5 - double(int(cfg->window_size * 0.5))) - 1.5;
    }
    // TriggeringProcessor::setSettings, StreamPreprocessing.h:46-53
    h.windowChanged = std::ceil(cfg->window_size) != std::ceil(h.windowSize) ? 1 : 0;
    h.windowSize = cfg->window_size;
    h.threshold = cfg->trigger_threshold;
    h.hysteresis = cfg->trigger_hysteresis;
    SGZ_HIP(hipMemcpy(s->d_state, &h, sizeof(h), hipMemcpyHostToDevice));
    s->size = size; s->backCap = backCap;
    // calculateTriggerIndices, OscilloscopeParameters.h:491-507
    const size_t idx = size_t(std::llround(cfg->trigger_channel - 1));
    s->trigSeparate = uint32_t(std::min<size_t>(C - 1, idx));
    s->trigPair = uint32_t(std::min<size_t>(C / 4, idx) * 2);
    s->envelopeCoeff = float(std::exp(-1.0 / (cfg->envelope_window * cfg->sample_rate)));   // OscilloscopeDSP.inl:448
    if (!s->d_col) SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_col), sizeof(ColourParams)));
    SGZ_HIP(hipStreamSynchronize(s->stream));                                               // (no ingest launch is reading the old copy)
    SGZ_HIP(hipMemcpy(s->d_col, &s->col, sizeof(ColourParams), hipMemcpyHostToDevice));
    s->cfg = *cfg;
    return SGZ_OK;
}

extern "C" {

#ifdef SGZ_DEBUG
sgz_status sgz_debug_ingest_clocks(unsigned long long *out8)
{
    SGZ_HIP(hipDeviceSynchronize());
    SGZ_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ingestClk), 8 * sizeof(unsigned long long)));
    return SGZ_OK;
}
#endif

sgz_status sgz_scope_create(const sgz_scope_config *cfg, sgz_scope **out)
{
    if (!cfg || !out) return fail(SGZ_EINVAL, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(SGZ_EHIP, "no HIP device visible (libsgz has no CPU fallback)");
    sgz_scope *s = new (std::nothrow) sgz_scope();
    if (!s) return fail(SGZ_ENOMEM, "out of memory");
    const sgz_status st = scopeSetup(s, cfg, true);
    if (st != SGZ_OK) { scopeFree(s); return st; }
    *out = s;
    return SGZ_OK;
}

void sgz_scope_destroy(sgz_scope *s) { scopeFree(s); }

static sgz_status scopeSync(sgz_scope *s);

sgz_status sgz_scope_configure(sgz_scope *s, const sgz_scope_config *cfg)
{
    if (!s || !cfg) return fail(SGZ_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;            // the audio already taken goes through the old configuration
    return scopeSetup(s, cfg, false);
}

// the open batch -> GPU: one staged copy, one launch of the ingest kernel over its blocks (caller holds the batch flag; count > 0)
static sgz_status scopeSubmit(sgz_scope *s, const PeakParams *peak = nullptr)
{
    sgz_status st;
    const float *fetchFrom; uint32_t floats;
    const float *d_batch = s->batch.upload(s->stream, &st, &fetchFrom, &floats);
    if (!d_batch) return st;
    IngestParams prm{};
    prm.batchHost = fetchFrom; prm.batchFloats = floats;
    prm.st = s->d_state; prm.peaks = s->d_peaks; prm.swapList = s->d_swaps;
    prm.batch = d_batch; prm.numBlocks = s->batch.count; prm.channels = s->cfg.num_channels;
    for (uint32_t b = 0; b < s->batch.count; ++b) { prm.blockOff[b] = s->batch.off[b]; prm.blockLen[b] = s->batch.len[b]; }
    prm.front = s->d_front; prm.size = s->size; prm.back = s->d_back; prm.backCap = s->backCap;
    prm.triggerMode = s->cfg.trigger_mode; prm.oscMode = s->cfg.channel_mode; prm.envMode = s->cfg.envelope_mode;
    prm.trigSeparate = s->trigSeparate; prm.trigPair = s->trigPair; prm.envelopeCoeff = s->envelopeCoeff;
    prm.colours = s->cfg.colour_by_frequency ? 1u : 0u;
    if (peak) { prm.doPeak = 1u; prm.peak = *peak; }
    hipLaunchKernelGGL(scopeIngestKernel, dim3(1), dim3(1024), 0, s->stream, prm, s->d_col);
    SGZ_HIP(hipGetLastError());
    return s->batch.commit(s->stream);
}

// the handle's GPU side for rt_lockfree.hpp's hand-over protocol (batchPush / batchSync / batchFlushAll: one block behind the ones already
// staged, parked blocks behind the open batch's in order, flush on read -- the same code the ThreadSanitizer harness runs on a mock GPU)
namespace {
struct ScopeIngestSide {
    sgz_scope *s;
    BatchCore &batch() { return s->batch; }
    Backlog &backlog() { return s->backlog; }
    sgz_status submit() { return scopeSubmit(s); }
    sgz_status slotReady() { return s->batch.slotReady(); }
    bool gpuIdle() { return s->batch.idle(); }
    void waitGpu() { (void)hipStreamSynchronize(s->stream); }
    bool deferSubmit() { return s->deferSubmit.load(std::memory_order_relaxed); }
};
}  // namespace

// consumer side (flush on read): what waits in the host FIFO and in the open batch goes to the GPU in front of the caller's own work
static sgz_status scopeSync(sgz_scope *s)
{
    ScopeIngestSide side{s};
    return batchSync(side);
}

sgz_status sgz_scope_push(sgz_scope *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples)
{
    if (!s || !planar) return fail(SGZ_EINVAL, "null argument");
    std::unique_lock<std::mutex> lk(s->mu, std::try_to_lock);      // never waits: a reconfiguration in progress drops the block
    if (!lk.owns_lock()) { s->busy++; return SGZ_BUSY; }
    if (num_channels != s->cfg.num_channels) return fail(SGZ_EINVAL, "num_channels differs from the configuration");
    if (nsamples == 0) return SGZ_OK;                              // audioEntryPoint returns at once (:403-404)
    if (nsamples > s->maxBlock) return fail(SGZ_EINVAL, "block longer than sgz_scope_config::max_block");
    // never waits: the render thread is submitting the open batch right now -> the block waits its turn in the host FIFO, like one
    // the GPU is not ready for (rt_common.hpp Backlog); SGZ_BUSY = that FIFO is full
    // (SGZ_RT_OPT_PARK_PUSHES: every block takes that way -- the tests' handle on a race that timing alone produces)
    ScopeIngestSide side{s};
    const sgz_status st = batchPush(side, planar, num_channels, nsamples, s->parkPushes.load(std::memory_order_relaxed));
    if (st == SGZ_BUSY) s->busy++;
    return st;
}

void *sgz_scope_stream(sgz_scope *s) { return s ? s->stream : nullptr; }

sgz_status sgz_scope_set_option(sgz_scope *s, uint32_t option, uint64_t value)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    if (option == SGZ_RT_OPT_PARK_PUSHES) { s->parkPushes.store(value != 0, std::memory_order_relaxed); return SGZ_OK; }
    if (option != SGZ_RT_OPT_DEFER_SUBMIT) return fail(SGZ_EINVAL, "unknown scope option");
    s->deferSubmit.store(value != 0, std::memory_order_relaxed);
    return SGZ_OK;
}

sgz_status sgz_scope_flush(sgz_scope *s)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    ScopeIngestSide side{s};
    return batchFlushAll(side);                                       // (this call may wait: it is not the audio thread's)
}

sgz_status sgz_scope_peak_filter(sgz_scope *s, double delta_time, uint32_t lanes, double *auto_gain)
{
    if (!s || lanes == 0 || (lanes & (lanes - 1))) return fail(SGZ_EINVAL, "bad argument");
    // coeff = pow(exp(-lanes / (envelopeWindow * sampleRate)), numSamples * dt), OscilloscopeDSP.inl:745-747
    const bool spectral = s->cfg.trigger_mode == SGZ_TRIG_SPECTRAL;
    const uint32_t numSamples = spectral ? uint32_t(s->trig.ring_size) : s->size;          // audioData.getSize()
    const double power = double(numSamples) * delta_time;
    const double coeff = std::pow(std::exp(-double(lanes) / (s->cfg.envelope_window * s->cfg.sample_rate)), power);
    PeakParams prm{s->d_state, s->d_front, s->size, s->cfg.num_channels, s->cfg.channel_mode, lanes, coeff, spectral ? numSamples : 0u};
    // flush on read: the blocks that wait in the open batch come first -- and the filter rides on their launch (one workgroup either
    // way: a launch and the gap in front of it less per rendered frame)
    ScopeIngestSide side{s};
    s->batch.lock();
    const sgz_status tb = batchTakeBacklog(side, false);
    const bool fused = tb == SGZ_OK && s->batch.count != 0;
    const sgz_status sy = tb != SGZ_OK ? tb : fused ? scopeSubmit(s, &prm) : SGZ_OK;
    s->batch.unlock();
    if (sy != SGZ_OK) return sy;
    if (!fused) {
        hipLaunchKernelGGL(scopePeakKernel, dim3(1), dim3(1024), 0, s->stream, prm);
        SGZ_HIP(hipGetLastError());
    }
    if (auto_gain) {
        SGZ_HIP(hipMemcpyAsync(auto_gain, reinterpret_cast<const char *>(s->d_state) + offsetof(ScopeDev, autoGain), sizeof(double),
                               hipMemcpyDeviceToHost, s->stream));
        SGZ_HIP(hipStreamSynchronize(s->stream));
    }
    return SGZ_OK;
}

sgz_status sgz_scope_gains(sgz_scope *s, double *envelope_gain, float *envelopes)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;            // (flush on read: the blocks that wait in the open batch come first)
    ScopeDev h;
    SGZ_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    if (envelope_gain) *envelope_gain = h.envelopeGain;
    if (envelopes) std::memcpy(envelopes, h.envelope, sizeof(float) * s->cfg.num_channels);
    return SGZ_OK;
}

sgz_status sgz_scope_front(sgz_scope *s, uint32_t channel, float *out, uint32_t *size, uint32_t *cursor)
{
    if (!s || channel >= s->cfg.num_channels) return fail(SGZ_EINVAL, "bad argument");
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;            // (flush on read: the blocks that wait in the open batch come first)
    ScopeDev h;
    if (out) SGZ_HIP(hipMemcpyAsync(out, s->d_front + size_t(channel) * s->size, size_t(s->size) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    if (size) *size = s->size;
    if (cursor) *cursor = h.frontCursor;
    return SGZ_OK;
}

sgz_status sgz_scope_front_colours(sgz_scope *s, uint32_t channel, uint32_t aux, uint8_t *out)
{
    if (!s || !out || channel >= s->cfg.num_channels || aux > 1) return fail(SGZ_EINVAL, "bad argument");
    if (!s->cfg.colour_by_frequency) return fail(SGZ_EINVAL, "colour_by_frequency is off");
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;            // (flush on read: the blocks that wait in the open batch come first)
    const size_t plane = size_t(aux ? s->cfg.num_channels : 0u) + channel;
    SGZ_HIP(hipMemcpyAsync(out, s->col.front + plane * s->size, size_t(s->size) * 4, hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

#ifdef SGZ_DEBUG
// (debug builds: the spectral trigger's median ring, 8 x (index, value, offset) as doubles)
sgz_status sgz_scope_debug_median(sgz_scope *s, double out[24])
{
    SpectralDev h;
    SGZ_HIP(hipMemcpyAsync(&h, s->d_spectral, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    for (int i = 0; i < 8; ++i) { out[3 * i] = double(h.median[i].index); out[3 * i + 1] = h.median[i].value; out[3 * i + 2] = h.median[i].offset; }
    return SGZ_OK;
}
#endif

sgz_status sgz_scope_debug_state(sgz_scope *s, uint64_t out[8])
{
    if (!s || !out) return fail(SGZ_EINVAL, "null argument");
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;            // (flush on read: the blocks that wait in the open batch come first)
    ScopeDev h;
    SGZ_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    out[0] = h.frontOrigin; out[1] = h.bufferedSamples; out[2] = h.oldPeak; out[3] = h.currentPeak;
    out[4] = h.steadyClock; out[5] = h.qCount; out[6] = uint64_t(h.isWorkingOnPeak); out[7] = h.swaps;
    return SGZ_OK;
}

sgz_status sgz_scope_set_transport(sgz_scope *s, int64_t position_in_samples)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    s->transport.store(position_in_samples, std::memory_order_relaxed);
    return SGZ_OK;
}

size_t sgz_scope_vertex_count(const sgz_scope *s, const sgz_scope_view *view)
{
    if (!s || !view || view->width < 2 || !(view->right > view->left)) return 0;
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;
    return scopeVertexCount(v, s->cfg.interpolation, s->cfg.trigger_mode, s->trig.cycle_samples);
}

sgz_status sgz_scope_analyse(sgz_scope *s, uint32_t evaluator, uint32_t channel, sgz_trigger_state *out)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;            // (flush on read: the blocks that wait in the open batch come first)
    if (s->cfg.trigger_mode == SGZ_TRIG_SPECTRAL) {
        const uint32_t C = s->cfg.num_channels;
        uint32_t chA, chB, evalMode;
        switch (evaluator) {
        case SGZ_OSC_LEFT: chA = chB = channel; evalMode = 0; break;
        case SGZ_OSC_RIGHT: chA = chB = channel + 1; evalMode = 0; break;
        case SGZ_OSC_MID: chA = channel; chB = channel + 1; evalMode = 1; break;
        case SGZ_OSC_SIDE: chA = channel; chB = channel + 1; evalMode = 2; break;
        default: return fail(SGZ_EINVAL, "evaluator: SGZ_OSC_LEFT / RIGHT / MID / SIDE");
        }
        if (chA >= C || chB >= C) return fail(SGZ_EINVAL, "channel out of range");
        SpectralParams prm{};
        prm.st = s->d_spectral;
        prm.ringA = s->d_front + size_t(chA) * s->size; prm.ringB = s->d_front + size_t(chB) * s->size;
        prm.evalMode = evalMode; prm.cap = s->size;
        prm.d_cursor = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s->d_state) + offsetof(ScopeDev, frontCursor));
        prm.windowSize = s->cfg.window_size; prm.sampleRate = s->cfg.sample_rate;
        prm.threshold = s->cfg.trigger_threshold; prm.hysteresis = s->cfg.trigger_hysteresis;
        prm.customFrequency = s->cfg.custom_trigger ? s->cfg.custom_trigger_frequency : 0.0;
        prm.phaseOffsetDeg = s->cfg.trigger_phase_offset;
        prm.quarterSemitone = std::pow(2, 0.25 / 12.0) - 1;                                  // OscilloscopeDSP.inl:126
        prm.tw = s->d_tw;
        hipLaunchKernelGGL(scopeSpectralKernel, dim3(1), dim3(1024), 2 * 8192 * sizeof(double), s->stream, prm);
        SGZ_HIP(hipGetLastError());
        SpectralDev h;
        SGZ_HIP(hipMemcpyAsync(&h, s->d_spectral, sizeof(h), hipMemcpyDeviceToHost, s->stream));
        SGZ_HIP(hipStreamSynchronize(s->stream));
        s->trig.record_index = h.record.index; s->trig.record_value = h.record.value; s->trig.record_offset = h.record.offset;
        s->trig.fundamental = h.fundamental; s->trig.cycle_samples = h.cycleSamples; s->trig.sample_offset = h.sampleOffset;
        s->trig.phase = h.phase; s->trig.ring_size = h.ringSize;
    }
    if (out) *out = s->trig;
    return SGZ_OK;
}

// one evaluator's vertex stream into DEVICE buffers (the handle's own, or the caller's mapped VBO); *points = vertices written
// what an evaluator reads: SampleColourEvaluator<OscChannels::...>, SampleColourEvaluators.h: Left / Right one channel, Mid / Side 0.5 (l +- r)
struct StripSource { const float *ringA, *ringB; const uint32_t *colRing; uint32_t evalMode, key; };
static sgz_status scopeStripSource(sgz_scope *s, uint32_t evaluator, uint32_t channel, bool wantColours, StripSource *out)
{
    const uint32_t C = s->cfg.num_channels;
    uint32_t chA, chB, evalMode, colourCh;
    switch (evaluator) {
    case SGZ_OSC_LEFT: chA = chB = channel; evalMode = 0; colourCh = channel; break;
    case SGZ_OSC_RIGHT: chA = chB = channel + 1; evalMode = 0; colourCh = channel + 1; break;
    case SGZ_OSC_MID: chA = channel; chB = channel + 1; evalMode = 1; colourCh = channel; break;
    case SGZ_OSC_SIDE: chA = channel; chB = channel + 1; evalMode = 2; colourCh = channel + 1; break;
    default: return fail(SGZ_EINVAL, "evaluator: SGZ_OSC_LEFT / RIGHT / MID / SIDE");
    }
    if (chA >= C || chB >= C) return fail(SGZ_EINVAL, "channel out of range");
    uint32_t key;
    std::memcpy(&key, s->cfg.colours[colourCh], 4);                    // evaluator.getDefaultKey()
    // colourChannelsByFrequency: Left / Right read colourData of their channel, Mid / Side auxColourData (SampleColourEvaluators.h:64,183)
    const uint32_t *colRing = nullptr;
    if (s->cfg.colour_by_frequency && wantColours)
        colRing = s->col.front + size_t((evalMode == 0 ? 0u : C) + colourCh) * s->size;
    *out = StripSource{s->d_front + size_t(chA) * s->size, s->d_front + size_t(chB) * s->size, colRing, evalMode, key};
    return SGZ_OK;
}

static sgz_status scopeVerticesInto(sgz_scope *s, const sgz_scope_view *view, uint32_t evaluator, uint32_t channel, float *d_xyz,
                                    uint32_t *d_rgba, size_t capacity, size_t *points)
{
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;            // (flush on read: the blocks that wait in the open batch come first)
    StripSource src;
    if (sgz_status st = scopeStripSource(s, evaluator, channel, d_rgba != nullptr, &src); st != SGZ_OK) return st;
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;                               // state.effectiveWindowSize is the stream's
    SGZ_HIP(launchScopeVertices(v, s->cfg.trigger_mode, s->cfg.interpolation, src.ringA, src.ringB, src.evalMode, uint32_t(s->trig.ring_size), s->size,
                                reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s->d_state) + offsetof(ScopeDev, frontCursor)),
                                s->trig.cycle_samples, s->trig.sample_offset, s->transport.load(std::memory_order_relaxed), src.key, src.colRing, d_xyz,
                                d_rgba, capacity, points, s->stream));
    return SGZ_OK;
}

// two strips of one view: Lanczos strips share their tap weights in ONE launch (scope_vector.hip scopeWaveLanczosKernel<2>); false: not Lanczos
static sgz_status scopeVerticesPairInto(sgz_scope *s, const sgz_scope_view *view, const uint32_t *evaluators, const uint32_t *channels,
                                        float *const d_xyz[2], uint32_t *const d_rgba[2], size_t capacity, size_t *points, bool *done)
{
    *done = false;
    if (sgz_status sy = scopeSync(s); sy != SGZ_OK) return sy;
    StripSource a, b;
    if (sgz_status st = scopeStripSource(s, evaluators[0], channels[0], d_rgba[0] != nullptr, &a); st != SGZ_OK) return st;
    if (sgz_status st = scopeStripSource(s, evaluators[1], channels[1], d_rgba[1] != nullptr, &b); st != SGZ_OK) return st;
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;
    const float *ra[2] = {a.ringA, b.ringA}, *rb[2] = {a.ringB, b.ringB};
    const uint32_t em[2] = {a.evalMode, b.evalMode}, key[2] = {a.key, b.key};
    const uint32_t *cr[2] = {a.colRing, b.colRing};
    hipError_t e = hipSuccess;
    *done = launchScopeLanczosPair(v, s->cfg.trigger_mode, s->cfg.interpolation, ra, rb, em, uint32_t(s->trig.ring_size), s->size,
                                   reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s->d_state) + offsetof(ScopeDev, frontCursor)),
                                   s->trig.cycle_samples, s->trig.sample_offset, s->transport.load(std::memory_order_relaxed), key, cr, d_xyz, d_rgba,
                                   capacity, points, s->stream, &e);
    if (*done) SGZ_HIP(e);
    return SGZ_OK;
}

sgz_status sgz_scope_vertices(sgz_scope *s, const sgz_scope_view *view, uint32_t evaluator, uint32_t channel, float *xyz, uint8_t *rgba,
                              uint32_t *count)
{
    if (!s || !view || !xyz || !count) return fail(SGZ_EINVAL, "null argument");
    if (view->width < 2 || !(view->right > view->left)) return fail(SGZ_EINVAL, "bad view");
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;
    const size_t need = scopeVertexCount(v, s->cfg.interpolation, s->cfg.trigger_mode, s->trig.cycle_samples);
    if (need > *count) { *count = uint32_t(need); return fail(SGZ_EINVAL, "vertex buffer too small (count holds the required size)"); }
    if (s->vertexCap < need) {
        if (s->d_xyz) (void)hipFree(s->d_xyz);
        if (s->d_rgba) (void)hipFree(s->d_rgba);
        if (s->h_out) (void)hipHostFree(s->h_out);
        s->d_xyz = nullptr; s->d_rgba = nullptr; s->h_out = nullptr; s->vertexCap = 0;
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_xyz), need * 3 * sizeof(float)));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_rgba), need * sizeof(uint32_t)));
        SGZ_HIP(hipHostMalloc(&s->h_out, need * 16, hipHostMallocDefault));
        s->vertexCap = need;
    }
    size_t points = 0;
    // pinned, device-mapped destinations: the vertex kernel writes them itself (2.4 MB per evaluator at cfg3: the DMA copy behind the
    // kernel was most of a rendered frame's GPU time)
    void *mx = mappedDevicePointer(xyz, s->stream), *mc = rgba ? mappedDevicePointer(rgba, s->stream) : nullptr;
    if (mx && (!rgba || mc)) {
        const sgz_status sd = scopeVerticesInto(s, view, evaluator, channel, static_cast<float *>(mx), static_cast<uint32_t *>(mc), need, &points);
        if (sd != SGZ_OK) return sd;
        SGZ_HIP(hipStreamSynchronize(s->stream));
        *count = uint32_t(points);
        return SGZ_OK;
    }
    const sgz_status st = scopeVerticesInto(s, view, evaluator, channel, s->d_xyz, rgba ? s->d_rgba : nullptr, need, &points);
    if (st != SGZ_OK) return st;
    if (sgz_status rb = readBack(xyz, s->d_xyz, points * 3 * sizeof(float), rgba, s->d_rgba, points * sizeof(uint32_t), s->h_out, s->stream);
        rb != SGZ_OK) return rb;
    *count = uint32_t(points);
    return SGZ_OK;
}

sgz_status sgz_scope_vertices_all(sgz_scope *s, const sgz_scope_view *view, uint32_t items, const uint32_t *evaluators, const uint32_t *channels,
                                  float *const *xyz, uint8_t *const *rgba, uint32_t *counts)
{
    if (!s || !view || !evaluators || !channels || !xyz || !counts) return fail(SGZ_EINVAL, "null argument");
    if (view->width < 2 || !(view->right > view->left)) return fail(SGZ_EINVAL, "bad view");
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;
    const size_t need = scopeVertexCount(v, s->cfg.interpolation, s->cfg.trigger_mode, s->trig.cycle_samples);
    bool small = false, direct = true;
    for (uint32_t k = 0; k < items; ++k) {
        if (!xyz[k]) return fail(SGZ_EINVAL, "null argument");
        if (need > counts[k]) { counts[k] = uint32_t(need); small = true; }
        direct = direct && mappedDevicePointer(xyz[k], s->stream) && (!rgba || !rgba[k] || mappedDevicePointer(rgba[k], s->stream));
    }
    if (small) return fail(SGZ_EINVAL, "vertex buffer too small (counts hold the required size)");
    if (!direct) {                                            // a pageable buffer among them: item by item through the bounce buffer
        for (uint32_t k = 0; k < items; ++k) {
            const sgz_status st = sgz_scope_vertices(s, view, evaluators[k], channels[k], xyz[k], rgba ? rgba[k] : nullptr, &counts[k]);
            if (st != SGZ_OK) return st;
        }
        return SGZ_OK;
    }
    for (uint32_t k = 0; k < items; ++k) {
        size_t points = 0;
        if (k + 1 < items) {                                  // two strips at a time when they are Lanczos strips (shared tap weights, one launch)
            float *px[2] = {static_cast<float *>(mappedDevicePointer(xyz[k], s->stream)), static_cast<float *>(mappedDevicePointer(xyz[k + 1], s->stream))};
            uint32_t *pc[2] = {rgba && rgba[k] ? static_cast<uint32_t *>(mappedDevicePointer(rgba[k], s->stream)) : nullptr,
                               rgba && rgba[k + 1] ? static_cast<uint32_t *>(mappedDevicePointer(rgba[k + 1], s->stream)) : nullptr};
            bool done = false;
            const sgz_status sp = scopeVerticesPairInto(s, view, evaluators + k, channels + k, px, pc, need, &points, &done);
            if (sp != SGZ_OK) { (void)hipStreamSynchronize(s->stream); return sp; }
            if (done) { counts[k] = counts[k + 1] = uint32_t(points); ++k; continue; }
        }
        const sgz_status st = scopeVerticesInto(s, view, evaluators[k], channels[k], static_cast<float *>(mappedDevicePointer(xyz[k], s->stream)),
                                                rgba && rgba[k] ? static_cast<uint32_t *>(mappedDevicePointer(rgba[k], s->stream)) : nullptr, need, &points);
        if (st != SGZ_OK) { (void)hipStreamSynchronize(s->stream); return st; }
        counts[k] = uint32_t(points);
    }
    SGZ_HIP(hipStreamSynchronize(s->stream));
    return SGZ_OK;
}

sgz_status sgz_scope_vertices_device(sgz_scope *s, const sgz_scope_view *view, uint32_t evaluator, uint32_t channel, float *d_xyz,
                                     uint8_t *d_rgba, uint32_t *count)
{
    if (!s || !view || !d_xyz || !count) return fail(SGZ_EINVAL, "null argument");
    if (view->width < 2 || !(view->right > view->left)) return fail(SGZ_EINVAL, "bad view");
    if ((reinterpret_cast<uintptr_t>(d_xyz) & 3) || (reinterpret_cast<uintptr_t>(d_rgba) & 3)) return fail(SGZ_EINVAL, "4-byte aligned buffers");
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;
    const size_t need = scopeVertexCount(v, s->cfg.interpolation, s->cfg.trigger_mode, s->trig.cycle_samples);
    if (need > *count) { *count = uint32_t(need); return fail(SGZ_EINVAL, "vertex buffer too small (count holds the required size)"); }
    size_t points = 0;
    const sgz_status st = scopeVerticesInto(s, view, evaluator, channel, d_xyz, reinterpret_cast<uint32_t *>(d_rgba), need, &points);
    if (st != SGZ_OK) return st;
    SGZ_HIP(hipStreamSynchronize(s->stream));                 // the vertices are in place when the call returns
    *count = uint32_t(points);
    return SGZ_OK;
}

}  // extern "C"
