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
#include <hip/hip_runtime.h>

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
hipError_t launchScopeVertices(const sgz_scope_view &view, uint32_t triggerMode, uint32_t interpolation, const float *ringA,
                               const float *ringB, uint32_t evalMode, uint32_t size, const uint32_t *d_cursor, uint32_t rgba,
                               float *d_xyz, uint32_t *d_rgba, size_t capacity, size_t *points, hipStream_t stream);
size_t scopeVertexCount(const sgz_scope_view &view, uint32_t interpolation);
}

namespace {

constexpr uint32_t kPeakCap = 1u << 16;       // pending triggers (std::queue<std::uint64_t> peaks)
constexpr uint32_t kMaxSwaps = kPeakCap + 8;  // one swap pops one trigger
constexpr uint32_t kMaxCh = 64;

struct Swap { unsigned long long src; unsigned int len; unsigned int pad; };

struct ScopeDev {
    // TriggeringProcessor (StreamPreprocessing.h:210-226)
    double threshold, windowSize, state;
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

struct IngestParams {
    ScopeDev *st;
    unsigned long long *peaks;                // [kPeakCap]
    Swap *swapList;                           // [kMaxSwaps]
    const float *block;                       // [channels][n]
    uint32_t n, channels;
    float *front; uint32_t size;              // [channels][size]
    float *back; uint32_t backCap;            // [channels][backCap], power of two
    uint32_t triggerMode, oscMode, envMode;
    uint32_t trigSeparate, trigPair;
    float envelopeCoeff;
};

__device__ __forceinline__ long long waveInclusiveMax(long long v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long u = __shfl_up(v, o);
        if (lane >= o && u > v) v = u;
    }
    return v;
}
// exclusive prefix-max over the workgroup's threads in thread order, seeded with `init`; *total = max over everything (and init)
__device__ __forceinline__ long long blockExclusiveMax(long long v, long long init, long long *lds, long long *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const long long inc = waveInclusiveMax(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    long long base = init;
    for (int w = 0; w < wave; ++w) base = lds[w] > base ? lds[w] : base;
    long long tot = init;
    for (int w = 0; w < waves; ++w) tot = lds[w] > tot ? lds[w] : tot;
    long long prev = __shfl_up(inc, 1);
    if (lane == 0) prev = -(1ll << 62);
    __syncthreads();
    *total = tot;
    return prev > base ? prev : base;
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

__device__ __forceinline__ double trigSample(uint32_t mode, const float *a, const float *b, uint32_t i)
{
    if (mode == SGZ_OSC_MID) return double(0.5f * (a[i] + b[i]));          // OscilloscopeDSP.inl:371-376
    if (mode == SGZ_OSC_SIDE) return double(0.5f * (a[i] - b[i]));         // :377-382
    return double(a[i]);
}

__device__ __forceinline__ unsigned long long minU64(unsigned long long a, unsigned long long b) { return a < b ? a : b; }

__global__ void __launch_bounds__(1024) scopeIngestKernel(const IngestParams prm)
{
    __shared__ long long sScan[16];
    __shared__ unsigned int sSum[16];
    __shared__ unsigned int sNumSwaps, sLastStart, sLastLen, sCursor0;
    __shared__ unsigned long long sWritten0;
    ScopeDev *st = prm.st;
    const uint32_t n = prm.n, C = prm.channels;
    const int tid = threadIdx.x, T = blockDim.x;
    const unsigned long long playhead = st->playhead;

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

    // ---- A: ZeroCrossingProcessor over the block (executeSamplingWindows, OscilloscopeDSP.inl:311-385)
    if (prm.triggerMode == 4u && C >= 2) {
        uint32_t localMode = prm.oscMode, pair = prm.trigPair;
        if (localMode == SGZ_OSC_MIDSIDE) { localMode = SGZ_OSC_MID; pair = prm.trigSeparate & ~1u; }     // :340-352
        const float *a, *b;
        if (localMode == SGZ_OSC_RIGHT) a = b = prm.block + size_t(pair + 1) * n;
        else if (localMode == SGZ_OSC_LEFT) a = b = prm.block + size_t(pair) * n;
        else if (localMode == SGZ_OSC_SEPARATE) a = b = prm.block + size_t(prm.trigSeparate) * n;
        else { a = prm.block + size_t(pair) * n; b = a + n; }
        const double threshold = st->threshold, prevState = st->state;
        const int armedIn = st->isPeakHold;
        const unsigned long long originIn = st->crossOrigin;
        const uint32_t seg = (n + T - 1) / T;
        const uint32_t i0 = min(n, uint32_t(tid) * seg), i1 = min(n, i0 + seg);
        // arm_i = (s_i > 0 && s_{i-1} < 0); fire_i <=> s_i > threshold && lastArm(i) > lastThr(i-1)  (virtual indices: an arm
        // inherited from the previous block sits at -1, "no arm" at -3, "no threshold crossing yet" at -2)
        long long segArm = -(1ll << 62), segThr = -(1ll << 62);
        for (uint32_t i = i0; i < i1; ++i) {
            const double s = trigSample(localMode, a, b, i);
            const double prev = i ? trigSample(localMode, a, b, i - 1) : prevState;
            if (s > 0 && prev < 0) segArm = i;
            if (s > threshold) segThr = i;
        }
        long long totArm, totThr;
        const long long inArm = blockExclusiveMax(segArm, armedIn ? -1 : -3, sScan, &totArm);
        const long long inThr = blockExclusiveMax(segThr, -2, sScan, &totThr);
        long long la = inArm, lt = inThr;
        unsigned int fires = 0;
        for (uint32_t i = i0; i < i1; ++i) {
            const double s = trigSample(localMode, a, b, i);
            const double prev = i ? trigSample(localMode, a, b, i - 1) : prevState;
            if (s > 0 && prev < 0) la = i;
            if (s > threshold) { if (la > lt) ++fires; lt = i; }
        }
        unsigned int totalFires;
        unsigned int pos = blockExclusiveSum(fires, sSum, &totalFires);
        const unsigned int q0 = st->qHead, qc0 = st->qCount;
        la = inArm; lt = inThr;
        for (uint32_t i = i0; i < i1; ++i) {
            const double s = trigSample(localMode, a, b, i);
            const double prev = i ? trigSample(localMode, a, b, i - 1) : prevState;
            if (s > 0 && prev < 0) la = i;
            if (s > threshold) {
                if (la > lt) {
                    if (qc0 + pos < kPeakCap) prm.peaks[(q0 + qc0 + pos) % kPeakCap] = (la == -1) ? originIn : playhead + (unsigned long long)la;
                    ++pos;
                }
                lt = i;
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

    // ---- B: processMutating's automaton (thread 0) -> swap list
    if (tid == 0) {
        unsigned int numSwaps = 0, lastStart = 0, lastLen = n;
        sCursor0 = st->frontCursor;
        sWritten0 = st->written;
        if (prm.triggerMode == 4u) {
            unsigned long long bufferedSamples = st->bufferedSamples, frontOrigin = st->frontOrigin, steadyClock = st->steadyClock;
            unsigned long long oldPeak = st->oldPeak, currentPeak = st->currentPeak;
            unsigned int qHead = st->qHead, qCount = st->qCount;
            int isWorkingOnPeak = st->isWorkingOnPeak;
            unsigned long long numSamples = n, consumed = 0;
            if (frontOrigin + bufferedSamples < steadyClock) { frontOrigin = steadyClock; bufferedSamples = 0; }   // :81-85
            const double ceilingSize = ceil(st->windowSize);
            const double halfSize = ceilingSize / 2;
            auto processIntoBackBuffer = [&](unsigned long long samples) {                                     // :90-105
                lastStart = (unsigned int)consumed; lastLen = (unsigned int)samples;
                consumed += samples;
                numSamples -= samples;
                const unsigned long long oldSamples = bufferedSamples;
                steadyClock += samples;
                bufferedSamples += samples;
                bufferedSamples = minU64(bufferedSamples, (unsigned long long)(ceilingSize + 1));
                frontOrigin += (oldSamples + samples) - bufferedSamples;
            };
            if (ceilingSize == 0 && qCount) qCount = 0;                                                        // :107-110
            while (numSamples != 0) {
                if (!qCount) { processIntoBackBuffer(numSamples); break; }
                else if (!isWorkingOnPeak) {
                    isWorkingOnPeak = 1;
                    const unsigned long long nextPeak = prm.peaks[qHead];
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
                        prm.swapList[numSwaps].src = (sWritten0 + consumed) - bufferedSamples;
                        prm.swapList[numSwaps].len = (unsigned int)cappedSize;
                        ++numSwaps;
                    }
                    bufferedSamples -= minU64(bufferedSamples, cappedSize);
                    frontOrigin += cappedSize;
                    oldPeak = currentPeak;
                    isWorkingOnPeak = 0;
                    if (qCount) { qHead = (qHead + 1) % kPeakCap; qCount--; }
                    st->swaps++;
                }
            }
            st->bufferedSamples = bufferedSamples; st->frontOrigin = frontOrigin; st->steadyClock = steadyClock;
            st->oldPeak = oldPeak; st->currentPeak = currentPeak; st->qHead = qHead; st->qCount = qCount;
            st->isWorkingOnPeak = isWorkingOnPeak;
        }
        sNumSwaps = numSwaps; sLastStart = lastStart; sLastLen = lastLen;
    }
    __syncthreads();

    // ---- C: swaps (ZeroCrossing) or the block itself (None: audioProcessing straight into the front buffer), in order
    const uint32_t size = prm.size;
    uint32_t cursor = sCursor0;
    const unsigned long long written0 = sWritten0;
    auto appendFront = [&](unsigned long long src, uint32_t len) {
        // only the last `size` samples of a longer run survive in the ring
        const uint32_t skip = len > size ? len - size : 0;
        const uint32_t cur0 = uint32_t((cursor + skip) % size);
        const uint32_t m = len - skip;
        for (uint32_t e = tid; e < m * C; e += T) {
            const uint32_t c = e / m, i = e - c * m;
            const unsigned long long abs = src + skip + i;
            const float v = abs >= written0 ? prm.block[size_t(c) * n + uint32_t(abs - written0)]
                                            : prm.back[size_t(c) * prm.backCap + uint32_t(abs & (prm.backCap - 1))];
            uint32_t d = cur0 + i; if (d >= size) d -= size;
            prm.front[size_t(c) * size + d] = v;
        }
        cursor = uint32_t((cursor + len) % size);
        __syncthreads();
    };
    if (prm.triggerMode == 4u) {
        const uint32_t ns = sNumSwaps;
        for (uint32_t k = 0; k < ns; ++k) appendFront(prm.swapList[k].src, prm.swapList[k].len);
    } else appendFront(written0, n);

    // ---- D: the block goes into the back rings (ZeroCrossing only; absolute index mod backCap)
    if (prm.triggerMode == 4u) {
        const uint32_t keep = n > prm.backCap ? prm.backCap : n, first = n - keep;
        for (uint32_t e = tid; e < keep * C; e += T) {
            const uint32_t c = e / keep, i = first + (e - c * keep);
            prm.back[size_t(c) * prm.backCap + uint32_t((written0 + i) & (prm.backCap - 1))] = prm.block[size_t(c) * n + i];
        }
    }

    // ---- E: RMS envelope (audioProcessing, OscilloscopeDSP.inl:520-585, :676-693); wave 0, lane c = recurrence c
    if (prm.envMode != 0u && tid < 64) {
        const bool active = tid < int(C);
        const float k = prm.envelopeCoeff;
        const uint32_t mode = prm.oscMode;
        const uint32_t lastStart = sLastStart, lastLen = sLastLen;
        // channels 0 / 1 are stored back (:690-691) and so run through every audioProcessing call of the block; the others restart
        // from their stored envelope in every call, so only the last call's samples matter for them
        const uint32_t from = tid < 2 ? 0u : lastStart, to = lastStart + lastLen;
        float y = active ? st->envelope[tid] : 0.f;
        const float *b0 = prm.block, *b1 = prm.block + n, *bc = prm.block + size_t(active ? tid : 0) * n;
        bool own = false;
        if (mode == SGZ_OSC_SEPARATE) {
            own = true;
            if (active) for (uint32_t i = from; i < to; ++i) { const float s = bc[i] * bc[i]; y = s + k * (y - s); }
        } else if (mode == SGZ_OSC_MIDSIDE) {
            if (tid < 2) {
                own = true;
                for (uint32_t i = 0; i < to; ++i) {
                    const float l = b0[i], r = b1[i];
                    const float s = tid == 0 ? 0.5f * ((l + r) * (l + r)) : 0.5f * ((l - r) * (l - r));
                    y = s + k * (y - s);
                }
            }
        } else if (tid == 0) {
            own = true;
            const float *src = mode == SGZ_OSC_RIGHT ? b1 : b0;
            for (uint32_t i = 0; i < to; ++i) {
                float v;
                if (mode == SGZ_OSC_MID) v = 0.5f * (b0[i] + b1[i]);
                else if (mode == SGZ_OSC_SIDE) v = 0.5f * (b0[i] - b1[i]);
                else v = src[i];
                const float s = v * v;
                y = s + k * (y - s);
            }
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
    if (tid == 0) {
        st->frontCursor = cursor;
        if (prm.triggerMode == 4u) st->written = written0 + n;
        st->playhead = playhead + n;
    }
}

// ---- Oscilloscope::runPeakFilter on the front rings (memory order, last size mod lanes slots dropped), every channel mode
struct PeakParams {
    ScopeDev *st;
    const float *front; uint32_t size, channels, mode, lanes;
    double coeff;
};
__global__ void __launch_bounds__(1024) scopePeakKernel(const PeakParams prm)
{
    __shared__ float sL[kMaxCh][16], sR[16];
    const uint32_t C = prm.channels, size = prm.size;
    const uint32_t mode = C == 1 ? uint32_t(SGZ_OSC_LEFT) : prm.mode;
    const uint32_t stop = size - (size & (prm.lanes - 1));
    const int tid = threadIdx.x, wave = tid >> 6, waves = blockDim.x >> 6;
    auto blockMax = [&](float v, float *slot) {
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        if ((tid & 63) == 0) slot[wave] = v;
    };
    const float *L = prm.front, *R = prm.front + (C > 1 ? size : 0);
    if (mode <= SGZ_OSC_SIDE) {
        float m = 0.f;
        for (uint32_t i = tid; i < stop; i += blockDim.x) {
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
            for (uint32_t i = tid; i < stop; i += blockDim.x) m = fmaxf(fabsf(x[i]), m);
            blockMax(m, sL[c]);
        }
    } else {
        float ml = 0.f, mr = 0.f;
        for (uint32_t i = tid; i < stop; i += blockDim.x) {
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

}  // namespace

struct sgz_scope {
    sgz_scope_config cfg{};
    std::mutex mu;                    // configure (consumer thread) against push (producer: try_lock only, never waits)
    hipStream_t stream = nullptr;
    StageRing stage;
    ScopeDev *d_state = nullptr;
    unsigned long long *d_peaks = nullptr;
    Swap *d_swaps = nullptr;
    float *d_front = nullptr, *d_back = nullptr;
    uint32_t size = 0, backCap = 0;
    uint32_t trigSeparate = 0, trigPair = 0;
    float envelopeCoeff = 0.f;
    // vertex output (consumer side)
    float *d_xyz = nullptr; uint32_t *d_rgba = nullptr; size_t vertexCap = 0;
    void *h_out = nullptr; size_t hOutBytes = 0;          // pinned
    uint64_t busy = 0;
};

static void scopeFree(sgz_scope *s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    s->stage.release();
    for (void *p : {(void *)s->d_state, (void *)s->d_peaks, (void *)s->d_swaps, (void *)s->d_front, (void *)s->d_back, (void *)s->d_xyz,
                    (void *)s->d_rgba})
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
    if (c->trigger_mode != SGZ_TRIG_NONE && c->trigger_mode != SGZ_TRIG_ZERO_CROSSING)
        return fail(SGZ_EUNSUPPORTED, "trigger modes Spectral / Window / EnvelopeHold are not built (SURVEY 8(f) #3)");
    if (c->channel_mode > SGZ_OSC_MIDSIDE || c->envelope_mode > SGZ_ENV_PEAK_DECAY) return fail(SGZ_EINVAL, "enum value");
    if (c->interpolation != SGZ_SUBSAMPLE_LINEAR && c->interpolation != SGZ_SUBSAMPLE_LANCZOS)
        return fail(SGZ_EUNSUPPORTED, "sub-sample interpolation: Linear or Lanczos");
    if (!std::isfinite(c->trigger_threshold) || !std::isfinite(c->trigger_channel) || !(c->trigger_channel >= 1)) return fail(SGZ_EINVAL, "trigger");
    if (!std::isfinite(c->envelope_window) || c->envelope_window < 0) return fail(SGZ_EINVAL, "envelope_window");
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
    const uint32_t size = uint32_t(std::ceil(cfg->window_size + 1));                 // ChannelData.h:121
    uint32_t backCap = 1; while (backCap < size) backCap <<= 1;
    const uint32_t maxBlock = cfg->max_block ? cfg->max_block : 8192u;
    const bool realloc = fresh || C != s->cfg.num_channels || size != s->size || maxBlock != s->stage.maxBlock;
    ScopeDev h{};
    if (!fresh) SGZ_HIP(hipMemcpy(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost));
    if (realloc) {
        for (void **p : {(void **)&s->d_front, (void **)&s->d_back}) if (*p) { (void)hipFree(*p); *p = nullptr; }
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_front), size_t(C) * size * sizeof(float)));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_back), size_t(C) * backCap * sizeof(float)));
        SGZ_HIP(hipMemset(s->d_front, 0, size_t(C) * size * sizeof(float)));
        SGZ_HIP(hipMemset(s->d_back, 0, size_t(C) * backCap * sizeof(float)));
        if ((st = s->stage.init(C, maxBlock)) != SGZ_OK) return st;
        if (!s->d_state) {
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_state), sizeof(ScopeDev)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_peaks), size_t(kPeakCap) * sizeof(unsigned long long)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_swaps), size_t(kMaxSwaps) * sizeof(Swap)));
        }
        h.frontCursor = 0; h.written = 0;
    }
    // TriggeringProcessor::setSettings, StreamPreprocessing.h:46-53
    h.windowChanged = std::ceil(cfg->window_size) != std::ceil(h.windowSize) ? 1 : 0;
    h.windowSize = cfg->window_size;
    h.threshold = cfg->trigger_threshold;
    SGZ_HIP(hipMemcpy(s->d_state, &h, sizeof(h), hipMemcpyHostToDevice));
    s->size = size; s->backCap = backCap;
    // calculateTriggerIndices, OscilloscopeParameters.h:491-507
    const size_t idx = size_t(std::llround(cfg->trigger_channel - 1));
    s->trigSeparate = uint32_t(std::min<size_t>(C - 1, idx));
    s->trigPair = uint32_t(std::min<size_t>(C / 4, idx) * 2);
    s->envelopeCoeff = float(std::exp(-1.0 / (cfg->envelope_window * cfg->sample_rate)));   // OscilloscopeDSP.inl:448
    s->cfg = *cfg;
    return SGZ_OK;
}

extern "C" {

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

sgz_status sgz_scope_configure(sgz_scope *s, const sgz_scope_config *cfg)
{
    if (!s || !cfg) return fail(SGZ_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    return scopeSetup(s, cfg, false);
}

sgz_status sgz_scope_push(sgz_scope *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples)
{
    if (!s || !planar) return fail(SGZ_EINVAL, "null argument");
    std::unique_lock<std::mutex> lk(s->mu, std::try_to_lock);      // never waits: a reconfiguration in progress drops the block
    if (!lk.owns_lock()) { s->busy++; return SGZ_BUSY; }
    if (num_channels != s->cfg.num_channels) return fail(SGZ_EINVAL, "num_channels differs from the configuration");
    if (nsamples == 0) return SGZ_OK;                              // audioEntryPoint returns at once (:403-404)
    if (nsamples > s->stage.maxBlock) return fail(SGZ_EINVAL, "block longer than sgz_scope_config::max_block");
    sgz_status st;
    const float *d_block = s->stage.stage(planar, nsamples, s->stream, &st);
    if (!d_block) { if (st == SGZ_BUSY) s->busy++; return st; }
    IngestParams prm{};
    prm.st = s->d_state; prm.peaks = s->d_peaks; prm.swapList = s->d_swaps;
    prm.block = d_block; prm.n = nsamples; prm.channels = num_channels;
    prm.front = s->d_front; prm.size = s->size; prm.back = s->d_back; prm.backCap = s->backCap;
    prm.triggerMode = s->cfg.trigger_mode; prm.oscMode = s->cfg.channel_mode; prm.envMode = s->cfg.envelope_mode;
    prm.trigSeparate = s->trigSeparate; prm.trigPair = s->trigPair; prm.envelopeCoeff = s->envelopeCoeff;
    hipLaunchKernelGGL(scopeIngestKernel, dim3(1), dim3(1024), 0, s->stream, prm);
    SGZ_HIP(hipGetLastError());
    return s->stage.commit(s->stream);
}

sgz_status sgz_scope_peak_filter(sgz_scope *s, double delta_time, uint32_t lanes, double *auto_gain)
{
    if (!s || lanes == 0 || (lanes & (lanes - 1))) return fail(SGZ_EINVAL, "bad argument");
    // coeff = pow(exp(-lanes / (envelopeWindow * sampleRate)), numSamples * dt), OscilloscopeDSP.inl:745-747
    const double power = double(s->size) * delta_time;
    const double coeff = std::pow(std::exp(-double(lanes) / (s->cfg.envelope_window * s->cfg.sample_rate)), power);
    PeakParams prm{s->d_state, s->d_front, s->size, s->cfg.num_channels, s->cfg.channel_mode, lanes, coeff};
    hipLaunchKernelGGL(scopePeakKernel, dim3(1), dim3(1024), 0, s->stream, prm);
    SGZ_HIP(hipGetLastError());
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
    ScopeDev h;
    if (out) SGZ_HIP(hipMemcpyAsync(out, s->d_front + size_t(channel) * s->size, size_t(s->size) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    if (size) *size = s->size;
    if (cursor) *cursor = h.frontCursor;
    return SGZ_OK;
}

sgz_status sgz_scope_debug_state(sgz_scope *s, uint64_t out[8])
{
    if (!s || !out) return fail(SGZ_EINVAL, "null argument");
    ScopeDev h;
    SGZ_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    out[0] = h.frontOrigin; out[1] = h.bufferedSamples; out[2] = h.oldPeak; out[3] = h.currentPeak;
    out[4] = h.steadyClock; out[5] = h.qCount; out[6] = uint64_t(h.isWorkingOnPeak); out[7] = h.swaps;
    return SGZ_OK;
}

size_t sgz_scope_vertex_count(const sgz_scope *s, const sgz_scope_view *view)
{
    if (!s || !view || view->width < 2 || !(view->right > view->left)) return 0;
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;
    return scopeVertexCount(v, s->cfg.interpolation);
}

sgz_status sgz_scope_vertices(sgz_scope *s, const sgz_scope_view *view, uint32_t evaluator, uint32_t channel, float *xyz, uint8_t *rgba,
                              uint32_t *count)
{
    if (!s || !view || !xyz || !count) return fail(SGZ_EINVAL, "null argument");
    if (view->width < 2 || !(view->right > view->left)) return fail(SGZ_EINVAL, "bad view");
    const uint32_t C = s->cfg.num_channels;
    // SampleColourEvaluator<OscChannels::...>, SampleColourEvaluators.h: Left / Right read one channel, Mid / Side 0.5 (l +- r)
    uint32_t chA, chB, evalMode, colourCh;
    switch (evaluator) {
    case SGZ_OSC_LEFT: chA = chB = channel; evalMode = 0; colourCh = channel; break;
    case SGZ_OSC_RIGHT: chA = chB = channel + 1; evalMode = 0; colourCh = channel + 1; break;
    case SGZ_OSC_MID: chA = channel; chB = channel + 1; evalMode = 1; colourCh = channel; break;
    case SGZ_OSC_SIDE: chA = channel; chB = channel + 1; evalMode = 2; colourCh = channel + 1; break;
    default: return fail(SGZ_EINVAL, "evaluator: SGZ_OSC_LEFT / RIGHT / MID / SIDE");
    }
    if (chA >= C || chB >= C) return fail(SGZ_EINVAL, "channel out of range");
    sgz_scope_view v = *view;
    v.window_size = s->cfg.window_size;                               // state.effectiveWindowSize is the stream's
    const size_t need = scopeVertexCount(v, s->cfg.interpolation);
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
    uint32_t key;
    std::memcpy(&key, s->cfg.colours[colourCh], 4);                    // evaluator.getDefaultKey()
    size_t points = 0;
    SGZ_HIP(launchScopeVertices(v, s->cfg.trigger_mode, s->cfg.interpolation, s->d_front + size_t(chA) * s->size,
                                s->d_front + size_t(chB) * s->size, evalMode, s->size,
                                reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s->d_state) + offsetof(ScopeDev, frontCursor)),
                                key, s->d_xyz, rgba ? s->d_rgba : nullptr, need, &points, s->stream));
    float *hx = static_cast<float *>(s->h_out);
    uint32_t *hc = reinterpret_cast<uint32_t *>(hx + need * 3);
    SGZ_HIP(hipMemcpyAsync(hx, s->d_xyz, points * 3 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    if (rgba) SGZ_HIP(hipMemcpyAsync(hc, s->d_rgba, points * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    std::memcpy(xyz, hx, points * 3 * sizeof(float));
    if (rgba) std::memcpy(rgba, hc, points * 4);
    *count = uint32_t(points);
    return SGZ_OK;
}

}  // extern "C"
