// vector_stream.hip -- the Vectorscope's real-time handle (sgz_vector_*): audio history ring, the audio thread's one-pole
// filters and the polar plot's vertex / colour stream, all resident in HBM.  gfx950 only.
//
// Replaces VectorScope::Processor::onStreamAudio -> audioProcessing (Source/Vectorscope/Vectorscope.cpp:268-392) on the audio
// thread -- plus the cpl::AudioStream history the renderer reads (a CLIFOStream per channel) -- and on the render thread
// VectorScope::runPeakFilter (VectorscopeRendering.cpp:826-889) and drawPolarPlot (:500-746) for every channel pair.
//
// Layout: ring [channels][size], size = the audio history window in samples, one write cursor.  A render sees the ring as the
// reference's AudioBufferView does: section 0 = memory [cursor, size), section 1 = [0, cursor) (getItIndex / getItRange).
// One push = one staged copy + one launch (ring append + the eight one-pole recurrences + the RMS epilogue); push never waits.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cmath>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>

#include "rt_common.hpp"
#include "fade_chain.hpp"

#pragma clang fp contract(off)

using namespace sgz;

namespace {

struct VecDev {
    float env[2], bal[2][2], phase[2];        // FilterStates (Vectorscope.h:97-111): envelope, balance[Slow/Fast][L/R], phase[Slow/Fast]
    double envelopeGain;                      // Processor::envelopeGain (relaxed_atomic<double>, starts at 1)
    unsigned int cursor, pad;
    unsigned long long written;
};

struct VecIngest {
    VecDev *st;
    const float *batchHost; uint32_t batchFloats;             // the pinned slot the blocks are fetched from first (rt_common.hpp batchFetch), or null
    const float *batch; uint32_t numBlocks, channels;         // the staged blocks back to back: block b = [channels][blockLen[b]] at batch + blockOff[b]
    uint32_t blockOff[BatchRing::kMaxBlocks], blockLen[BatchRing::kMaxBlocks];
    float *ring; uint32_t size;
    uint32_t lanes, envMode;
    float envelope, pole0, pole1;
};

__global__ void __launch_bounds__(256) vectorIngestKernel(const VecIngest prm)
{
    constexpr int kTile = 2048;
    __shared__ float sL[kTile], sR[kTile], sP[kTile];
    VecDev *st = prm.st;
    const int tid = threadIdx.x;
    const uint32_t size = prm.size, C = prm.channels;
    // one launch takes every block that was waiting (rt_common.hpp BatchRing), one after the other with the host's block boundaries:
    // audioProcessing drops the SIMD tail of EVERY callback (Vectorscope.cpp:292)
    // (the block table goes through LDS: a run-time subscript into the by-value argument struct would move the struct to scratch)
    __shared__ uint32_t sBlockOff[BatchRing::kMaxBlocks], sBlockLen[BatchRing::kMaxBlocks];
#pragma unroll
    for (uint32_t b = 0; b < BatchRing::kMaxBlocks; ++b)
        if (tid == int(b)) { sBlockOff[b] = prm.blockOff[b]; sBlockLen[b] = prm.blockLen[b]; }
    __syncthreads();
    batchFetch(prm.batchHost, const_cast<float *>(prm.batch), prm.batchFloats, tid, 256);
    for (uint32_t blockIndex = 0; blockIndex < prm.numBlocks; ++blockIndex) {
    const float *const blk = prm.batch + sBlockOff[blockIndex];
    const uint32_t n = sBlockLen[blockIndex];
    const uint32_t cursor0 = st->cursor;
    // ring append (only the newest `size` samples of a longer block survive)
    const uint32_t skip = n > size ? n - size : 0, m = n - skip;
    const uint32_t cur0 = (cursor0 + skip) % size;
    for (uint32_t e = tid; e < m * C; e += blockDim.x) {
        const uint32_t c = e / m, i = e - c * m;
        uint32_t d = cur0 + i; if (d >= size) d -= size;
        prm.ring[size_t(c) * size + d] = blk[size_t(c) * n + skip + i];
    }
    // audioProcessing on channels 0 / 1 (Vectorscope.cpp:268-377): wave 0; lane k < 8 owns one recurrence:
    // 0,1 envelope L/R; 2,3 slow balance L/R; 4,5 fast balance L/R; 6 slow phase; 7 fast phase
    const uint32_t np = n - (n & (prm.lanes - 1));                        // :292, the SIMD tail of the block is dropped
    // The eight recurrences' inputs -- l^2, r^2 and cos(2 atan(vY / vX)) of every sample -- do not depend on the recurrences: all 256
    // threads evaluate them for a tile of kTile samples into LDS, then lanes 0 .. 7 of wave 0 walk the tile, sixteen inputs per LDS
    // round trip.  (Until round 4 wave 0 alternated the two in 64-sample steps with two wave barriers each: the other three waves
    // idled and the chain stopped for an atan and a cos every 64 steps -- 13 us per 512-sample block.)
    float y = 0.f, a = 0.f;
    if (tid < 8) {
        const float *sp = reinterpret_cast<const float *>(st);
        y = sp[tid];
        a = tid < 2 ? prm.envelope : ((tid == 2 || tid == 3 || tid == 6) ? prm.pole0 : prm.pole1);
    }
    const int sel = (tid == 0 || tid == 2 || tid == 4) ? 0 : ((tid == 1 || tid == 3 || tid == 5) ? 1 : 2);
    for (uint32_t base = 0; base < np; base += kTile) {
        const uint32_t cnt = min(uint32_t(kTile), np - base);
        const float *L = blk, *R = blk + n;
        for (uint32_t k = tid; k < cnt; k += blockDim.x) {
            const float l = L[base + k], r = R[base + k];
            const float mReal = -0.70710678118654752440f, mImag = 0.70710678118654752440f;
            const float vX = l * mReal - r * mImag;                    // :303
            const float vY = r * mImag + l * mReal;                    // :304
            const float radians = atanf(vY / vX);                      // :310
            const float ang = (vX == 0.f && vY == 0.f) ? 0.78539816339744830962f : radians;   // :311
            sP[k] = cosf(ang * 2.0f);                                  // :316
            sL[k] = l * l; sR[k] = r * r;                              // :323-324
        }
        __syncthreads();
        if (tid < 8) {
            const float *src = sel == 0 ? sL : (sel == 1 ? sR : sP);
            uint32_t k0 = 0;
            for (; k0 + 16 <= cnt; k0 += 16) {                         // whole batches: straight-line chain, no predicate per step
                float xs[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) xs[j] = src[k0 + uint32_t(j)];
#pragma unroll
                for (int j = 0; j < 16; ++j) { const float x = xs[j]; y = x + a * (y - x); }                       // :327-342
            }
            for (; k0 < cnt; ++k0) { const float x = src[k0]; y = x + a * (y - x); }
        }
        __syncthreads();
    }
    if (tid < 64) {
        const int lane = tid;
        // :346-376: the envelope filters are stored (and the gain refreshed) only in the RMS mode; balance and phase always
        const float e0 = __shfl(y, 0), e1 = __shfl(y, 1);
        if (lane < 8 && (lane >= 2 || prm.envMode == 1u)) reinterpret_cast<float *>(st)[lane] = y;
        if (lane == 0 && prm.envMode == 1u) {
            const double currentEnvelope = 1.0 / double(fmaxf(__builtin_sqrtf(e0), __builtin_sqrtf(e1)));   // std::sqrt(T), T = float
            const double ax = fabs(currentEnvelope);
            if (ax >= 2.2250738585072014e-308 && ax < INFINITY) st->envelopeGain = currentEnvelope;         // std::isnormal
        }
    }
    __syncthreads();
    if (tid == 0) { st->cursor = (cursor0 + n) % size; st->written += n; }
    __syncthreads();                                                      // (the next block reads the cursor and the filter states)
    }
}

// VectorScope::runPeakFilter: memory-order maximum of |x| over channels 0 / 1 (the last size mod lanes slots dropped), then
// envelope = max(envelope * coeff, peak^2) in double, stored as float; envelopeGain = 1 / max sqrt, if normal
__global__ void __launch_bounds__(1024) vectorPeakKernel(VecDev *st, const float *ring, uint32_t size, uint32_t lanes, double coeff)
{
    __shared__ float sm[2][16];
    const uint32_t stop = size - (size & (lanes - 1));
    const int tid = threadIdx.x, wave = tid >> 6;
    float ml = 0.f, mr = 0.f;
    for (uint32_t i = tid; i < stop; i += blockDim.x) { ml = fmaxf(ml, fabsf(ring[i])); mr = fmaxf(mr, fabsf(ring[size + i])); }
    for (int o = 32; o > 0; o >>= 1) { ml = fmaxf(ml, __shfl_xor(ml, o)); mr = fmaxf(mr, __shfl_xor(mr, o)); }
    if ((tid & 63) == 0) { sm[0][wave] = ml; sm[1][wave] = mr; }
    __syncthreads();
    if (tid != 0) return;
    float hl = 0.f, hr = 0.f;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) { hl = fmaxf(hl, sm[0][w]); hr = fmaxf(hr, sm[1][w]); }
    const double highestLeft = hl, highestRight = hr;
    st->env[0] = float(fmax(double(st->env[0]) * coeff, highestLeft * highestLeft));       // :873-874
    st->env[1] = float(fmax(double(st->env[1]) * coeff, highestRight * highestRight));
    const double currentEnvelope = 1.0 / fmax(sqrt(double(st->env[0])), sqrt(double(st->env[1])));   // :876
    const double ax = fabs(currentEnvelope);
    if (ax >= 2.2250738585072014e-308 && ax < INFINITY) st->envelopeGain = currentEnvelope;
}

// The fade ramp is a running fp32 SIMD sum (vSampleFade += fadePerSample * V once per SIMD iteration, += fadePerSample *
// remainder after each section, VectorscopeRendering.cpp:528-543, :592, :634): sequential as written but a function of
// (size, cursor, lanes) only.  ramp[] = vSampleFade of every SIMD-body sample (in vertex order), tail[s] = outFade[V-1] as the
// scalar tail of section s sees it.  One thread per SIMD lane finds the chain's arithmetic progressions (fade_chain.hpp: one per
// binade the sum crosses), all threads evaluate; a chain with ties at every step is walked the reference's way (rampSequential).
__device__ void rampSequential(float *buf, uint32_t lane, uint32_t lanes, uint32_t size, uint32_t cursor, float *ramp, float *tail)
{
    const long V = long(lanes);
    const float fadePerSample = 1.0f / float(size);
    const float incr = fadePerSample * float(V);
    float f = fadePerSample * float(lane);                  // vSampleFade = outFade = fadePerSample * i
    float lastOut = f;                                       // outFade[lane] (only lane V-1's matters)
    size_t base = 0;
    const long chunk = (8192 / V) * V;                       // samples per LDS round: whole SIMD iterations
    for (int section = 0; section < 2; ++section) {
        const long n = section == 0 ? long(size - cursor) : long(cursor);
        const long body = n - V > 0 ? ((n - V + V - 1) / V) * V : 0;        // samples the SIMD loop (i < n - V; i += V) covers
        for (long i0 = 0; i0 < body; i0 += chunk) {
            const long m = body - i0 < chunk ? body - i0 : chunk;
            if (lane < lanes) {
                const int steps = int(m / V), stride = int(V);
                float *dst = buf + lane;
                float prev = f;
                for (int k = 0; k < steps; ++k) { *dst = f; prev = f; f += incr; dst += stride; }
                if (steps > 0) lastOut = prev - 1.0f;
            }
            __syncthreads();
            for (long e = lane; e < m; e += 256) ramp[base + size_t(i0 + e)] = buf[e];
            __syncthreads();
        }
        if (lane == lanes - 1) tail[section] = lastOut;
        const long remaining = n - body > 0 ? n - body : 0;
        f += fadePerSample * float(remaining);
        base += size_t(n);
    }
}

constexpr uint32_t kRampLanes = 16;                           // SIMD widths the progression form keeps tables for (wider: sequential)

// The first section's chain starts at fadePerSample * lane whatever the cursor is: its progressions are found ONCE per (size, lanes)
// (one more step than any section has: the value the walk would write next is what the second section starts from); n[lane] = -1:
// the table overflowed (ties at every step) and the per-frame kernel walks.
__global__ void __launch_bounds__(64) vectorRampTableKernel(uint32_t size, uint32_t lanes, FadeSeg *table /*[kRampLanes][kFadeSegs]*/, int *count /*[kRampLanes]*/)
{
    const uint32_t lane = threadIdx.x;
    if (lane >= lanes || lane >= kRampLanes) return;
    const float fadePerSample = 1.0f / float(size);
    float f = fadePerSample * float(lane), last = 0.f;
    int n = 0;
    const bool ok = fadeChainSegments(f, fadePerSample * float(lanes), size / lanes + 1u, table + size_t(lane) * kFadeSegs, n, last);
    count[lane] = ok ? n : -1;
}

// Per rendered frame: the second section's chain starts where the first one's ended plus the scalar tail's share -- a function of the
// cursor.  One thread per lane reads that value off the configured table and finds the second section's progressions (one to three
// where the ring is half full or more); frame[lane] / frameCount[lane] and tail[] are what the polar kernel evaluates per vertex.
// frameCount[0] = -1: some chain overflowed its table -- ramp[] was filled the reference's way and the polar kernel reads that.
__global__ void __launch_bounds__(256) vectorRampKernel(const VecDev *st, uint32_t size, uint32_t lanes, const FadeSeg *table, const int *tableCount,
                                                        FadeSeg *frame, int *frameCount, float *ramp, float *tail)
{
    __shared__ float buf[8192];
    __shared__ int overflow;
    const uint32_t lane = threadIdx.x;
    const uint32_t cursor = st->cursor;
    const long V = long(lanes);
    if (lane == 0) overflow = lanes > kRampLanes ? 1 : 0;
    __syncthreads();
    long bodyOf[2], nOf[2];
    for (int section = 0; section < 2; ++section) {
        nOf[section] = section == 0 ? long(size - cursor) : long(cursor);
        bodyOf[section] = nOf[section] - V > 0 ? ((nOf[section] - V + V - 1) / V) * V : 0;     // samples the SIMD loop (i < n - V; i += V) covers
    }
    if (lane < lanes && lanes <= kRampLanes) {
        const int n0 = tableCount[lane];
        if (n0 < 0) atomicExch(&overflow, 1);
        else {
            const float fadePerSample = 1.0f / float(size);
            const float incr = fadePerSample * float(V);
            const uint32_t steps0 = uint32_t(bodyOf[0] / V), steps1 = uint32_t(bodyOf[1] / V);
            const FadeSeg *mine = table + size_t(lane) * kFadeSegs;
            float lastOut = fadePerSample * float(lane), tails[2];
            if (steps0 > 0) lastOut = fadeChainValueAt(mine, n0, steps0 - 1u) - 1.0f;
            tails[0] = lastOut;
            float f = fadeChainValueAt(mine, n0, steps0);                          // what the walk would write next
            f += fadePerSample * float(nOf[0] - bodyOf[0] > 0 ? nOf[0] - bodyOf[0] : 0);
            float last = 0.f;
            int n = 0;
            const bool ok = fadeChainSegments(f, incr, steps1, frame + size_t(lane) * kFadeSegs, n, last);
            if (steps1 > 0) lastOut = last - 1.0f;
            tails[1] = lastOut;
            if (!ok) atomicExch(&overflow, 1);
            else {
                frameCount[lane] = n;
                if (lane == lanes - 1) { tail[0] = tails[0]; tail[1] = tails[1]; }
            }
        }
    }
    __syncthreads();
    if (overflow) {                                            // (uniform)
        rampSequential(buf, lane, lanes, size, cursor, ramp, tail);
        if (lane == 0) frameCount[0] = -1;
    }
}

struct PolarParams {
    const VecDev *st;
    const float *ring; uint32_t size, lanes, fade;
    const float *ramp, *tail;                                  // ramp: only read when frameCount[0] < 0 (vectorRampKernel's fallback)
    const FadeSeg *table, *frame; const int *tableCount, *frameCount;
    float3 *xyz, *rgb; size_t pairStride;                      // pair p's streams at xyz + p * pairStride (vertices)
    float colour[32][3];
    uint32_t firstPair;
};
// grid (vertices / 256, pairs of this call)
__global__ void __launch_bounds__(256) vectorPolarViewKernel(const PolarParams prm)
{
    const size_t v = size_t(blockIdx.x) * blockDim.x + threadIdx.x;     // vertex index: section 0 then section 1
    const uint32_t size = prm.size;
    if (v >= size) return;
    const uint32_t pair = prm.firstPair + blockIdx.y;
    const uint32_t cursor = prm.st->cursor;
    const long V = long(prm.lanes);
    const long n0 = long(size - cursor);
    const int section = long(v) < n0 ? 0 : 1;
    const long n = section == 0 ? n0 : long(cursor);
    const long i = section == 0 ? long(v) : long(v) - n0;               // index inside the section
    const uint32_t mem = section == 0 ? cursor + uint32_t(i) : uint32_t(i);
    const float l = prm.ring[size_t(2 * pair) * size + mem];
    const float r = prm.ring[size_t(2 * pair + 1) * size + mem];
    const float cosineRotation = -0.70710678118654752440f, sineRotation = 0.70710678118654752440f;   // :521-524
    const float length = fmaxf(fabsf(l), fabsf(r));                                                 // :563
    const float vY = l * cosineRotation - r * sineRotation;                                          // :566
    const float vX = l * sineRotation + r * cosineRotation;                                          // :567
    float angle = atanf(vX / vY);                                                                    // :576
    if (l == 0.f && r == 0.f) angle = 0.f;                                                           // :578
    float sx, cy;
    sincosf(angle, &sx, &cy);
    // SIMD body covers i < mainEnd = V * ceil((n - V) / V) (the loop `for (; i < n - V; i += V)`), the scalar tail the rest
    const long iters = n > V ? (n - V + V - 1) / V : 0;
    const long mainEnd = iters * V;
    const float fadePerSample = 1.0f / float(size);
    float z, cf;
    if (i < mainEnd) {
        float sf;
        if (prm.frameCount[0] >= 0) {                          // vSampleFade of SIMD iteration i / V, lane i % V: off the chain's progressions
            const uint32_t shift = uint32_t(__ffs(int(prm.lanes)) - 1), ln = uint32_t(i) & (prm.lanes - 1u), k = uint32_t(i) >> shift;   // (lanes: a power of two)
            sf = section == 0 ? fadeChainValueAt(prm.table + size_t(ln) * kFadeSegs, prm.tableCount[ln], k)
                              : fadeChainValueAt(prm.frame + size_t(ln) * kFadeSegs, prm.frameCount[ln], k);
        } else sf = prm.ramp[v];
        z = sf - 1.0f;                                       // outFade = vSampleFade - 1 (:592)
        cf = sf;                                             // colour * vSampleFade (:697-699)
    } else {
        z = prm.tail[section] - float(i - mainEnd) * fadePerSample;      // :600, :634
        cf = z + 1.0f;                                       // colour * (currentFade + 1) (:738)
    }
    const size_t at = size_t(blockIdx.y) * prm.pairStride + v;
    prm.xyz[at] = make_float3(sx * length, cy * length, z);
    const float *col = prm.colour[blockIdx.y];
    if (prm.rgb) prm.rgb[at] = prm.fade ? make_float3(col[0] * cf, col[1] * cf, col[2] * cf) : make_float3(col[0], col[1], col[2]);
}

}  // namespace

struct sgz_vector {
    sgz_vector_config cfg{};
    std::atomic<bool> deferSubmit{false};      // sgz_vector_set_option(SGZ_RT_OPT_DEFER_SUBMIT); read by whoever holds the batch flag
    std::atomic<bool> parkPushes{false};                  // ... (SGZ_RT_OPT_PARK_PUSHES): every push waits in the host FIFO for the next reader / flush
    std::mutex mu;
    hipStream_t stream = nullptr;
    BatchRing batch;                           // staged blocks waiting for their (one) ingest launch (rt_common.hpp)
    uint32_t maxBlock = 0;
    Backlog backlog;                           // blocks waiting for a staging slot (rt_common.hpp)
    VecDev *d_state = nullptr;
    float *d_ring = nullptr;
    uint32_t size = 0;
    float envelopeCoeff = 0.f, stereoCoeff = 0.f, pole1 = 0.f;
    float *d_ramp = nullptr, *d_tail = nullptr, *d_xyz = nullptr, *d_rgb = nullptr;
    FadeSeg *d_rampTable = nullptr; int *d_rampCount = nullptr;     // the first section's progressions, per (size, lanes); the second's, per frame, behind them
    uint32_t rampTableSize = 0, rampTableLanes = 0;
    void *h_out = nullptr;
    uint64_t busy = 0;
    // the fade ramp is a function of (size, cursor, lanes): one replay serves every pair of a rendered frame -- it is redone when a block
    // has been accepted since (pushes counts them; ~0 = never computed / reconfigured)
    std::atomic<uint64_t> pushes{0};
    uint64_t rampAt = ~0ull;
};

static void vectorFree(sgz_vector *s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    s->batch.release();
    s->backlog.release();
    for (void *p : {(void *)s->d_state, (void *)s->d_ring, (void *)s->d_ramp, (void *)s->d_tail, (void *)s->d_rampTable, (void *)s->d_rampCount, (void *)s->d_xyz, (void *)s->d_rgb})
        if (p) (void)hipFree(p);
    if (s->h_out) (void)hipHostFree(s->h_out);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

static sgz_status vectorSetup(sgz_vector *s, const sgz_vector_config *cfg, bool fresh)
{
    if (!(cfg->sample_rate >= 1) || !std::isfinite(cfg->sample_rate)) return fail(SGZ_EINVAL, "sample_rate");
    if (cfg->num_channels < 2 || (cfg->num_channels & 1) || cfg->num_channels > 64) return fail(SGZ_EINVAL, "num_channels must be even, 2..64");
    if (cfg->window_size < 1 || cfg->window_size > (1u << 26)) return fail(SGZ_EINVAL, "window_size");
    if (cfg->lanes == 0 || (cfg->lanes & (cfg->lanes - 1)) || cfg->lanes > 64) return fail(SGZ_EINVAL, "lanes must be a power of two <= 64");
    if (cfg->envelope_mode > SGZ_ENV_PEAK_DECAY) return fail(SGZ_EINVAL, "envelope_mode");
    if (!std::isfinite(cfg->envelope_window) || !std::isfinite(cfg->stereo_window) || cfg->envelope_window < 0 || cfg->stereo_window < 0)
        return fail(SGZ_EINVAL, "window times");
    if (!s->stream) SGZ_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    s->rampAt = ~0ull;
    const uint32_t C = cfg->num_channels, size = cfg->window_size;
    if (cfg->max_block > (1u << 17)) return fail(SGZ_EINVAL, "max_block above 131072 samples");
    const uint32_t maxBlock = cfg->max_block ? cfg->max_block : 8192u;
    const bool realloc = fresh || C != s->cfg.num_channels || size != s->size || maxBlock != s->maxBlock;
    if (realloc) {
        for (void **p : {(void **)&s->d_ring, (void **)&s->d_ramp, (void **)&s->d_xyz, (void **)&s->d_rgb}) if (*p) { (void)hipFree(*p); *p = nullptr; }
        if (s->h_out) { (void)hipHostFree(s->h_out); s->h_out = nullptr; }
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_ring), size_t(C) * size * sizeof(float)));
        SGZ_HIP(hipMemset(s->d_ring, 0, size_t(C) * size * sizeof(float)));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_ramp), size_t(size) * sizeof(float)));
        // (room for every pair's stream: sgz_vector_vertices_all renders them with one wait)
        const size_t pairsCap = std::max<size_t>(1, C / 2);
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_xyz), pairsCap * size * 3 * sizeof(float)));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_rgb), pairsCap * size * 3 * sizeof(float)));
        SGZ_HIP(hipHostMalloc(&s->h_out, pairsCap * size * 6 * sizeof(float), hipHostMallocDefault));
        sgz_status st = s->batch.init(C, std::max<uint32_t>(maxBlock, 8192u));     // a slot takes a whole batch: the blocks of a rendered frame and more
        s->maxBlock = maxBlock;
        if (st != SGZ_OK) return st;
        // one second of audio may wait for the GPU (at least 32 blocks)
        if (!s->backlog.init(backlogFloats(C, cfg->sample_rate, maxBlock))) return fail(SGZ_ENOMEM, "out of memory (push backlog)");
        if (!s->d_state) {
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_state), sizeof(VecDev)));
            SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_tail), 2 * sizeof(float)));
            VecDev h{};
            h.envelopeGain = 1.0;                          // Processor(): envelopeGain(1)
            SGZ_HIP(hipMemcpy(s->d_state, &h, sizeof(h), hipMemcpyHostToDevice));
        } else {
            VecDev h{};
            SGZ_HIP(hipMemcpy(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost));
            h.cursor = 0; h.written = 0;
            SGZ_HIP(hipMemcpy(s->d_state, &h, sizeof(h), hipMemcpyHostToDevice));
        }
    }
    s->size = size;
    // handleFlagUpdates, Vectorscope.cpp:201-202: relaxed_atomic<float> coefficients
    s->envelopeCoeff = float(std::exp(-1.0 / (cfg->envelope_window * cfg->sample_rate)));
    s->stereoCoeff = float(std::exp(-1.0 / (cfg->stereo_window * cfg->sample_rate)));
    s->pole1 = std::pow(s->stereoCoeff, 0.25f);           // secondStereoFilterSpeed(0.25f), Vectorscope.cpp:281
    s->cfg = *cfg;
    return SGZ_OK;
}

extern "C" {

sgz_status sgz_vector_create(const sgz_vector_config *cfg, sgz_vector **out)
{
    if (!cfg || !out) return fail(SGZ_EINVAL, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(SGZ_EHIP, "no HIP device visible (libsgz has no CPU fallback)");
    sgz_vector *s = new (std::nothrow) sgz_vector();
    if (!s) return fail(SGZ_ENOMEM, "out of memory");
    const sgz_status st = vectorSetup(s, cfg, true);
    if (st != SGZ_OK) { vectorFree(s); return st; }
    *out = s;
    return SGZ_OK;
}

void sgz_vector_destroy(sgz_vector *s) { vectorFree(s); }

static sgz_status vectorSync(sgz_vector *s);

sgz_status sgz_vector_configure(sgz_vector *s, const sgz_vector_config *cfg)
{
    if (!s || !cfg) return fail(SGZ_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (sgz_status sy = vectorSync(s); sy != SGZ_OK) return sy;           // the audio already taken goes through the old configuration
    return vectorSetup(s, cfg, false);
}

// the open batch -> GPU: one staged copy, one launch of the ingest kernel over its blocks (caller holds the batch flag; count > 0)
static sgz_status vectorSubmit(sgz_vector *s)
{
    sgz_status st;
    const float *fetchFrom; uint32_t floats;
    const float *d_batch = s->batch.upload(s->stream, &st, &fetchFrom, &floats);
    if (!d_batch) return st;
    VecIngest prm{};
    prm.batchHost = fetchFrom; prm.batchFloats = floats;
    prm.st = s->d_state; prm.batch = d_batch; prm.numBlocks = s->batch.count; prm.channels = s->cfg.num_channels;
    for (uint32_t b = 0; b < s->batch.count; ++b) { prm.blockOff[b] = s->batch.off[b]; prm.blockLen[b] = s->batch.len[b]; }
    prm.ring = s->d_ring; prm.size = s->size; prm.lanes = s->cfg.lanes; prm.envMode = s->cfg.envelope_mode;
    prm.envelope = s->envelopeCoeff; prm.pole0 = s->stereoCoeff; prm.pole1 = s->pole1;
    hipLaunchKernelGGL(vectorIngestKernel, dim3(1), dim3(256), 0, s->stream, prm);
    SGZ_HIP(hipGetLastError());
    s->pushes.fetch_add(s->batch.count, std::memory_order_release);
    return s->batch.commit(s->stream);
}

// the handle's GPU side for rt_lockfree.hpp's hand-over protocol (batchPush / batchSync / batchFlushAll; the ThreadSanitizer harness runs
// the same protocol code on a mock GPU)
namespace {
struct VectorIngestSide {
    sgz_vector *s;
    BatchCore &batch() { return s->batch; }
    Backlog &backlog() { return s->backlog; }
    sgz_status submit() { return vectorSubmit(s); }
    sgz_status slotReady() { return s->batch.slotReady(); }
    bool gpuIdle() { return s->batch.idle(); }
    void waitGpu() { (void)hipStreamSynchronize(s->stream); }
    bool deferSubmit() { return s->deferSubmit.load(std::memory_order_relaxed); }
};
}  // namespace

// consumer side (flush on read): what waits in the host FIFO and in the open batch goes to the GPU in front of the caller's own work
static sgz_status vectorSync(sgz_vector *s)
{
    VectorIngestSide side{s};
    return batchSync(side);
}

sgz_status sgz_vector_push(sgz_vector *s, const float *const *planar, uint32_t num_channels, uint32_t nsamples)
{
    if (!s || !planar) return fail(SGZ_EINVAL, "null argument");
    std::unique_lock<std::mutex> lk(s->mu, std::try_to_lock);
    if (!lk.owns_lock()) { s->busy++; return SGZ_BUSY; }
    if (num_channels != s->cfg.num_channels) return fail(SGZ_EINVAL, "num_channels differs from the configuration");
    if (nsamples == 0) return SGZ_OK;
    if (nsamples > s->maxBlock) return fail(SGZ_EINVAL, "block longer than sgz_vector_config::max_block");
    // never waits: the render thread is submitting the open batch right now -> the block waits its turn in the host FIFO, like one the
    // GPU is not ready for (rt_common.hpp Backlog); SGZ_BUSY = that FIFO is full
    // (SGZ_RT_OPT_PARK_PUSHES: every block takes that way -- the tests' handle on a race that timing alone produces)
    VectorIngestSide side{s};
    const sgz_status st = batchPush(side, planar, num_channels, nsamples, s->parkPushes.load(std::memory_order_relaxed));
    if (st == SGZ_BUSY) s->busy++;
    return st;
}

void *sgz_vector_stream(sgz_vector *s) { return s ? s->stream : nullptr; }

sgz_status sgz_vector_set_option(sgz_vector *s, uint32_t option, uint64_t value)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    if (option == SGZ_RT_OPT_PARK_PUSHES) { s->parkPushes.store(value != 0, std::memory_order_relaxed); return SGZ_OK; }
    if (option != SGZ_RT_OPT_DEFER_SUBMIT) return fail(SGZ_EINVAL, "unknown vector option");
    s->deferSubmit.store(value != 0, std::memory_order_relaxed);
    return SGZ_OK;
}

sgz_status sgz_vector_flush(sgz_vector *s)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    VectorIngestSide side{s};
    return batchFlushAll(side);                                       // (this call may wait: it is not the audio thread's)
}

sgz_status sgz_vector_peak_filter(sgz_vector *s, double delta_time, double *envelope_gain)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    if (sgz_status sy = vectorSync(s); sy != SGZ_OK) return sy;           // (flush on read: the blocks that wait in the open batch come first)
    // coeff = pow(envelopeCoeff, numSamples * openGLDeltaTime()), VectorscopeRendering.cpp:838-842 (envelopeCoeff is a float)
    const double coeff = std::pow(double(s->envelopeCoeff), double(s->size) * delta_time);
    hipLaunchKernelGGL(vectorPeakKernel, dim3(1), dim3(1024), 0, s->stream, s->d_state, s->d_ring, s->size, s->cfg.lanes, coeff);
    SGZ_HIP(hipGetLastError());
    if (envelope_gain) {
        SGZ_HIP(hipMemcpyAsync(envelope_gain, reinterpret_cast<const char *>(s->d_state) + offsetof(VecDev, envelopeGain), sizeof(double),
                               hipMemcpyDeviceToHost, s->stream));
        SGZ_HIP(hipStreamSynchronize(s->stream));
    }
    return SGZ_OK;
}

sgz_status sgz_vector_filters_get(sgz_vector *s, sgz_vector_filters *filters, double *envelope_gain)
{
    if (!s) return fail(SGZ_EINVAL, "null handle");
    if (sgz_status sy = vectorSync(s); sy != SGZ_OK) return sy;           // (flush on read: the blocks that wait in the open batch come first)
    VecDev h;
    SGZ_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    if (filters) {
        filters->env[0] = h.env[0]; filters->env[1] = h.env[1];
        for (int i = 0; i < 2; ++i) { filters->phase[i] = h.phase[i]; for (int j = 0; j < 2; ++j) filters->balance[i][j] = h.bal[i][j]; }
    }
    if (envelope_gain) *envelope_gain = h.envelopeGain;
    return SGZ_OK;
}

sgz_status sgz_vector_history(sgz_vector *s, uint32_t channel, float *out, uint32_t *size, uint32_t *cursor)
{
    if (!s || channel >= s->cfg.num_channels) return fail(SGZ_EINVAL, "bad argument");
    if (sgz_status sy = vectorSync(s); sy != SGZ_OK) return sy;           // (flush on read: the blocks that wait in the open batch come first)
    VecDev h;
    if (out) SGZ_HIP(hipMemcpyAsync(out, s->d_ring + size_t(channel) * s->size, size_t(s->size) * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipMemcpyAsync(&h, s->d_state, sizeof(h), hipMemcpyDeviceToHost, s->stream));
    SGZ_HIP(hipStreamSynchronize(s->stream));
    if (size) *size = s->size;
    if (cursor) *cursor = h.cursor;
    return SGZ_OK;
}

// the kernels of `pairs` consecutive pairs' vertex streams into DEVICE buffers (the handle's own, or the caller's mapped VBO): pair
// firstPair + p at d_xyz + p * 3 size floats
static sgz_status vectorVerticesInto(sgz_vector *s, uint32_t firstPair, uint32_t pairs, float *d_xyz, float *d_rgb)
{
    if (sgz_status sy = vectorSync(s); sy != SGZ_OK) return sy;           // (flush on read: the blocks that wait in the open batch come first)
    const uint32_t size = s->size;
    const uint64_t now = s->pushes.load(std::memory_order_acquire);
    if (!s->d_rampTable) {
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_rampTable), size_t(2) * kRampLanes * kFadeSegs * sizeof(FadeSeg)));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_rampCount), 2 * kRampLanes * sizeof(int)));
    }
    FadeSeg *frame = s->d_rampTable + size_t(kRampLanes) * kFadeSegs;
    int *frameCount = s->d_rampCount + kRampLanes;
    if (s->rampAt != now) {                                    // (a function of (size, cursor, lanes): once per rendered frame, not once per pair)
        if (s->rampTableSize != size || s->rampTableLanes != s->cfg.lanes) {
            hipLaunchKernelGGL(vectorRampTableKernel, dim3(1), dim3(64), 0, s->stream, size, s->cfg.lanes, s->d_rampTable, s->d_rampCount);
            s->rampTableSize = size; s->rampTableLanes = s->cfg.lanes;
        }
        hipLaunchKernelGGL(vectorRampKernel, dim3(1), dim3(256), 0, s->stream, s->d_state, size, s->cfg.lanes, s->d_rampTable, s->d_rampCount, frame, frameCount,
                           s->d_ramp, s->d_tail);
        s->rampAt = now;
    }
    PolarParams prm{};
    prm.st = s->d_state; prm.ring = s->d_ring; prm.size = size; prm.lanes = s->cfg.lanes; prm.fade = s->cfg.fade_history ? 1u : 0u;
    prm.ramp = s->d_ramp; prm.tail = s->d_tail;
    prm.table = s->d_rampTable; prm.tableCount = s->d_rampCount; prm.frame = frame; prm.frameCount = frameCount;
    prm.xyz = reinterpret_cast<float3 *>(d_xyz); prm.rgb = reinterpret_cast<float3 *>(d_rgb); prm.pairStride = size;
    for (uint32_t p = 0; p < pairs; ++p)
        for (int k = 0; k < 3; ++k) prm.colour[p][k] = s->cfg.colours[firstPair + p][k];
    prm.firstPair = firstPair;
    hipLaunchKernelGGL(vectorPolarViewKernel, dim3((size + 255) / 256, pairs), dim3(256), 0, s->stream, prm);
    SGZ_HIP(hipGetLastError());
    return SGZ_OK;
}

sgz_status sgz_vector_vertices(sgz_vector *s, uint32_t pair, float *xyz, float *rgb, uint32_t *count)
{
    if (!s || !xyz || !count) return fail(SGZ_EINVAL, "null argument");
    if (pair >= s->cfg.num_channels / 2) return fail(SGZ_EINVAL, "pair out of range");
    const uint32_t size = s->size;
    if (*count < size) { *count = size; return fail(SGZ_EINVAL, "vertex buffer too small (count holds the required size)"); }
    const sgz_status st = vectorVerticesInto(s, pair, 1, s->d_xyz, rgb ? s->d_rgb : nullptr);
    if (st != SGZ_OK) return st;
    const size_t bytes = size_t(size) * 3 * sizeof(float);
    if (sgz_status rb = readBack(xyz, s->d_xyz, bytes, rgb, s->d_rgb, bytes, s->h_out, s->stream); rb != SGZ_OK) return rb;
    *count = size;
    return SGZ_OK;
}

sgz_status sgz_vector_vertices_all(sgz_vector *s, float *xyz, float *rgb, uint32_t *count)
{
    if (!s || !xyz || !count) return fail(SGZ_EINVAL, "null argument");
    const uint32_t size = s->size, pairs = s->cfg.num_channels / 2;
    if (*count < size) { *count = size; return fail(SGZ_EINVAL, "vertex buffer too small (count holds the required size per pair)"); }
    const size_t per = size_t(size) * 3;
    // pinned, device-mapped destinations: the polar kernels write them themselves (no staging, no DMA copy behind the kernels)
    float *mx = static_cast<float *>(mappedDevicePointer(xyz, s->stream)), *mc = rgb ? static_cast<float *>(mappedDevicePointer(rgb, s->stream)) : nullptr;
    if (mx && (!rgb || mc)) {
        const sgz_status st = vectorVerticesInto(s, 0, pairs, mx, rgb ? mc : nullptr);       // every pair in one launch
        if (st != SGZ_OK) return st;
        SGZ_HIP(hipStreamSynchronize(s->stream));
        *count = size;
        return SGZ_OK;
    }
    if (const sgz_status st = vectorVerticesInto(s, 0, pairs, s->d_xyz, rgb ? s->d_rgb : nullptr); st != SGZ_OK) return st;
    const size_t bytes = pairs * per * sizeof(float);
    if (sgz_status rb = readBack(xyz, s->d_xyz, bytes, rgb, s->d_rgb, bytes, s->h_out, s->stream); rb != SGZ_OK) return rb;
    *count = size;
    return SGZ_OK;
}

sgz_status sgz_vector_vertices_device(sgz_vector *s, uint32_t pair, float *d_xyz, float *d_rgb, uint32_t *count)
{
    if (!s || !d_xyz || !count) return fail(SGZ_EINVAL, "null argument");
    if (pair >= s->cfg.num_channels / 2) return fail(SGZ_EINVAL, "pair out of range");
    if (*count < s->size) { *count = s->size; return fail(SGZ_EINVAL, "vertex buffer too small (count holds the required size)"); }
    const sgz_status st = vectorVerticesInto(s, pair, 1, d_xyz, d_rgb);
    if (st != SGZ_OK) return st;
    SGZ_HIP(hipStreamSynchronize(s->stream));                 // the vertices are in place when the call returns
    *count = s->size;
    return SGZ_OK;
}

}  // extern "C"
