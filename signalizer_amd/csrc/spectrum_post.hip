// spectrum_post.hip -- K_B: peak-decay recurrence across frames, log-dB scaling and the gradient colour
// map with additive blend over stereo pairs -> RGBA8 columns.  gfx950 only.
//
// Replaces TransformPair::mapAndTransformDFTFilters (Source/Spectrum/TransformDSP.inl:1299-1435) and
// AudioDispatcher::blendAndDispatchSpectrums (Source/Spectrum/SpectrumDSP.cpp:111-206).
//
// The recurrence  s_t = max(fl(s_{t-1} * pole), mag_t)  is sequential in the reference.  Here time is cut
// into chunks: because x -> fl(x * pole) is monotone, s_t = max(a_t, b_t) holds EXACTLY, where a_t is the
// chunk-local scan started from 0 and b_t is the carry (state at the end of the previous chunk) decayed by
// sequential fp32 multiplies.  So the chunked result is bit-identical to the sequential one; the same
// identity carries the state across GPUs in the multi-GPU time-chunk sharding.
//
// fp contraction is OFF in this file: every multiply/add must round exactly as the reference's scalar code.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "decay_body.hpp"      // fp contraction off from here on

namespace sgz {

// K_B1: chunk-end aggregates.  thread <-> (chunk, pair, side, pixel); both graphs per thread.
__global__ void __launch_bounds__(256) decayLocalKernel(const DecayParams prm)
{
    const size_t perChunk = size_t(prm.C) * prm.sides * prm.P;
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= perChunk * prm.numChunks) return;
    const uint32_t chunk = uint32_t(gid / perChunk);
    const size_t rem = gid - size_t(chunk) * perChunk;          // (pair, side, pixel) linear
    const uint32_t pixel = uint32_t(rem % prm.P);
    const uint32_t ps = uint32_t(rem / prm.P);                  // pair * sides + side
    const uint32_t pair = ps / prm.sides, side = ps - pair * prm.sides;

    float a[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const size_t si = ((size_t(pair) * G + k) * prm.P + pixel) * 2 + side;
        a[k] = (chunk == 0 && prm.stateIn) ? prm.stateIn[si] : 0.f;
        if (chunk == 0 && prm.stateStash) prm.stateStash[si] = a[k];   // the carry-in as the emit launch must see it (it overwrites state)
    }
    const long f0 = long(chunk) * kMaxChunk;
    const int len = int(min(long(kMaxChunk), prm.frames - f0));
    float mag[kMaxChunk];
#pragma unroll
    for (int t = 0; t < kMaxChunk; ++t)                         // independent loads first
        mag[t] = prm.mapped[size_t(f0 + (t < len ? t : 0)) * perChunk + rem] * prm.magScale;    // (x 1 is exact; 0.5: Phase, :1407)
#pragma unroll
    for (int t = 0; t < kMaxChunk; ++t) {
        if (t < len) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                a[k] = a[k] * prm.sc.pole[k];                   // states[i] *= pole, TransformDSP.inl:1336,:1370
                if (mag[t] > a[k]) a[k] = mag[t];               // :1338-1341
            }
        }
    }
#pragma unroll
    for (int k = 0; k < G; ++k)
        prm.agg[((size_t(chunk) * prm.C * prm.sides + ps) * G + k) * prm.P + pixel] = a[k];
}

// K_B1b: exact state at the END of every chunk, in place: agg[c] <- max(agg[c], decay^chunkLen(agg[c-1])) with the
// decay done by sequential fp32 multiplies.  thread <-> (pair, side, graph, pixel); sequential over chunks.
// OFF: uint32_t while the aggregates of the pass stay below 4 GB (every render of the BASELINE configs), size_t beyond.
template <typename OFF>
__global__ void __launch_bounds__(256) decayCarryKernel(const DecayParams prm)
{
    const size_t per = size_t(prm.C) * prm.sides * G * prm.P;
    const size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= per) return;
    const uint32_t k = uint32_t((e / prm.P) % G);
    const float pole = prm.sc.pole[k];
    float c = prm.agg[e];
    // The launch has C*sides*G*P threads -- one wave per SIMD at best -- and a lone wave issues one instruction per ~4.8
    // clocks whatever its dependencies (tools/ubench/valu.hip), so the fold costs its instruction count: per chunk one
    // 32-bit offset bump shared by the load and the store, kMaxChunk multiplies, a compare and a select.  B aggregates are
    // fetched together, the next batch while this one is folded.
    constexpr int B = 32;
    char *base = reinterpret_cast<char *>(prm.agg);
    const OFF stride = OFF(per * sizeof(float));
    OFF off = OFF(e * sizeof(float)) + stride;                       // chunk 1
    uint32_t d = 1;
    float a[B], nxt[B];
    auto fetch = [&](OFF o, float (&v)[B]) {
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = *reinterpret_cast<const float *>(base + (o + OFF(j) * stride));
    };
    if (d + B <= prm.numChunks) fetch(off, a);
    while (d + B <= prm.numChunks) {                                    // full batches: no per-chunk predicates
        const bool more = d + 2 * B <= prm.numChunks;
        if (more) fetch(off + B * stride, nxt);
#pragma unroll
        for (int j = 0; j < B; ++j) {
#pragma unroll
            for (int i = 0; i < kMaxChunk; ++i) c = c * pole;          // every chunk before the last is full
            if (a[j] > c) c = a[j];
            *reinterpret_cast<float *>(base + off) = c;
            off += stride;
        }
        d += B;
        if (more) {
#pragma unroll
            for (int j = 0; j < B; ++j) a[j] = nxt[j];
        }
    }
    for (; d < prm.numChunks; ++d) {                                    // tail
        const float v = *reinterpret_cast<const float *>(base + off);
#pragma unroll
        for (int i = 0; i < kMaxChunk; ++i) c = c * pole;
        if (v > c) c = v;
        *reinterpret_cast<float *>(base + off) = c;
        off += stride;
    }
}

// K_B1 + K_B1b in one launch for renders of at most kFusedChunks time chunks (cfg2: 44): a workgroup owns 16 (pair, side,
// pixel) entries for the whole time axis -- thread (chunk, entry) scans its chunk, the aggregates meet in LDS, 16 * G threads
// fold them sequentially (the same multiplies as decayCarryKernel), and everybody writes the carried states back.  One
// launch and one kernel boundary less than the two-kernel form, which long renders keep.
// 1024 threads = CH chunk slots x (1024 / CH) entries; CH = the power of two that holds the render's chunks (8 .. 64).
constexpr int kFusedChunks = 64;
template <int CH>
__global__ void __launch_bounds__(1024) decayLocalCarryKernel(const DecayParams prm)
{
    constexpr int kFusedEntries = 1024 / CH;
    __shared__ float aggS[CH][G][kFusedEntries];
    const size_t perChunk = size_t(prm.C) * prm.sides * prm.P;
    const int en = threadIdx.x & (kFusedEntries - 1);
    const uint32_t chunk = threadIdx.x / kFusedEntries;
    // colourOnly: the launch covers (pair, pixel) only -- side 0 of every pair -- and only LineMain is scanned
    const size_t lin = size_t(blockIdx.x) * kFusedEntries + en;
    const size_t nEntries = prm.colourOnly ? size_t(prm.C) * prm.P : perChunk;
    const uint32_t pixel = uint32_t(lin % prm.P);
    const uint32_t ps = prm.colourOnly ? uint32_t(lin / prm.P) * prm.sides : uint32_t(lin / prm.P);   // pair * sides + side
    const size_t rem = size_t(ps) * prm.P + pixel;             // (pair, side, pixel) linear
    const bool live = lin < nEntries && chunk < prm.numChunks;
    const uint32_t pair = ps / prm.sides, side = ps - pair * prm.sides;
    const int graphs = prm.colourOnly ? 1 : G;
    if (live) {
        float a[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const size_t si = ((size_t(pair) * G + k) * prm.P + pixel) * 2 + side;
            a[k] = (chunk == 0 && prm.stateIn) ? prm.stateIn[si] : 0.f;
            if (chunk == 0 && prm.stateStash) prm.stateStash[si] = a[k];   // the carry-in as the emit launch must see it (it overwrites state)
        }
        const long f0 = long(chunk) * kMaxChunk;
        const int len = int(min(long(kMaxChunk), prm.frames - f0));
        float mag[kMaxChunk];
#pragma unroll
        for (int t = 0; t < kMaxChunk; ++t) mag[t] = prm.mapped[size_t(f0 + (t < len ? t : 0)) * perChunk + rem] * prm.magScale;
#pragma unroll
        for (int t = 0; t < kMaxChunk; ++t) {
            if (t < len) {
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    if (k < graphs) {
                        a[k] = a[k] * prm.sc.pole[k];           // states[i] *= pole, TransformDSP.inl:1336,:1370
                        if (mag[t] > a[k]) a[k] = mag[t];       // :1338-1341
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < G; ++k) aggS[chunk][k][en] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < graphs * kFusedEntries) {
        const int k = threadIdx.x / kFusedEntries;
        const float pole = prm.sc.pole[k];
        float c = aggS[0][k][en];
        for (uint32_t d = 1; d < prm.numChunks; ++d) {
#pragma unroll
            for (int i = 0; i < kMaxChunk; ++i) c = c * pole;  // every chunk before the last is full
            const float v = aggS[d][k][en];
            if (v > c) c = v;
            aggS[d][k][en] = c;
        }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < G; ++k)
            if (k < graphs) prm.agg[((size_t(chunk) * prm.C * prm.sides + ps) * G + k) * prm.P + pixel] = aggS[chunk][k][en];
    }
}

// The whole of K_B as ONE launch for the commonest request: the colour column only (no line results, no state out), one pair, at
// most kFusedChunks time chunks.  A workgroup of 1024 threads owns PX = 4 pixels for the whole time axis:
//   1. threads (chunk, px) < 64 x 4 load their chunk's magnitudes (side 0), park them in LDS and scan the chunk from a zero carry;
//   2. 4 threads fold the chunk-end states with the same sequential multiplies as decayCarryKernel;
//   3. ALL threads share the frames x 4 emissions (replay <= 8 steps from the LDS magnitudes on the folded carry-in, dB map, colour).
// Nothing goes through HBM between the steps and there is one launch instead of two.
// The fused kernels' fold publishes how many chunk carries are final; the emitting waves wait for the one they need.
#define SGZ_PUBLISH(n) asm volatile("ds_write_b32 %0, %1" :: "v"(progressAddr), "v"(uint32_t(n)) : "memory")
__device__ __forceinline__ void awaitCarries(uint32_t progressAddr, uint32_t chunk)
{
    for (;;) {
        uint32_t seen;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(progressAddr) : "memory");
        if (seen >= chunk) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

// Workgroup -> group of PX adjacent pixels.  Workgroup b runs on XCD b % 8 (observed; a speed assumption only): XCD x takes the x-th
// eighth of the pixels, so that the 16-byte pieces its workgroups read of every magnitude row (and write of every image row) add up to
// whole cache lines inside ONE L2 -- with group = b, the eight workgroups that share a line sit on eight XCDs and each fetches it.
__device__ __forceinline__ uint32_t pixelGroup()
{
    const uint32_t b = blockIdx.x, n = gridDim.x;
    return (n & 7u) == 0 ? (b & 7u) * (n >> 3) + (b >> 3) : b;
}

template <int PX>
__global__ void __launch_bounds__(1024) decayColourFusedKernel(const DecayParams prm)
{
    __shared__ float aggS[kFusedChunks][PX];                    // chunk-end states of the zero-carry scans
    __shared__ float carryS[kFusedChunks][PX];                  // exact state at the end of chunk d (after the fold)
    __shared__ float magS[kFusedChunks * kMaxChunk][PX];
    __shared__ uint32_t progress;                               // carryS[d] is final for every d < progress
    __shared__ double logTab[16][2];
    stageLogfTable(logTab);
    const uint32_t progressAddr = uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)&progress));
    if (threadIdx.x == 0) progress = 0;
    const uint32_t tid = threadIdx.x;
    const size_t perFrame = size_t(prm.C) * prm.sides * prm.P;
    const float pole = prm.sc.pole[0];
    // 1a. every thread fetches its share of the frames x PX magnitudes (side 0 of the pair)
    const uint32_t items = uint32_t(prm.frames) * PX;
    // (every load of every round is requested before the first is waited for: as a loop with the late pixels' dependent reads inside, a
    // workgroup took two to four memory round trips here)
    {
        constexpr int ROUNDS = kFusedChunks * kMaxChunk * PX / 1024;
        float m[ROUNDS], nyRe[ROUNDS], nyIm[ROUNDS], best[ROUNDS];
        const bool lateGroup = prm.hasLate && pixelGroup() * PX + PX > prm.late.fixFrom0;     // (uniform)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = tid + 1024u * r, px = e % PX, f = e / PX;
            const uint32_t pixel = pixelGroup() * PX + px;
            const bool in = e < items && pixel < prm.P;
            m[r] = in ? prm.mapped[size_t(f) * perFrame + pixel] : 0.f;
            nyRe[r] = nyIm[r] = 0.f; best[r] = __builtin_inff();
            if (lateGroup && in && pixel >= prm.late.fixFrom0) {                               // (late_fix.hpp)
                nyRe[r] = prm.late.ny[2 * size_t(f)]; nyIm[r] = prm.late.ny[2 * size_t(f) + 1];
                best[r] = lateBestSquare(prm.late, long(f), 0, pixel);
            }
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = tid + 1024u * r, px = e % PX, f = e / PX;
            const uint32_t pixel = pixelGroup() * PX + px;
            float v = m[r] * prm.magScale;                                                     // (x 1 is exact)
            if (lateGroup && e < items && pixel >= prm.late.fixFrom0 && pixel < prm.P)
                v = lateNyquistPixel(prm.late, lateNyquistValue(prm.late, nyRe[r], nyIm[r]), best[r], v);
            magS[f][px] = v;
        }
    }
    __syncthreads();
    // 1b. zero-carry scan of every chunk
    if (tid < kFusedChunks * PX) {
        const uint32_t px = tid % PX, chunk = tid / PX;
        const uint32_t pixel = pixelGroup() * PX + px;
        const long f0 = long(chunk) * kMaxChunk;
        const int len = chunk < prm.numChunks ? int(min(long(kMaxChunk), prm.frames - f0)) : 0;
        float a = (pixel < prm.P && chunk == 0 && prm.stateIn) ? prm.stateIn[size_t(pixel) * 2] : 0.f;
#pragma unroll
        for (int t = 0; t < kMaxChunk; ++t)
            if (t < len) {
                const float m = magS[chunk * kMaxChunk + t][px];
                a = a * pole;                                   // states[i] *= pole, TransformDSP.inl:1336,:1370
                if (m > a) a = m;                               // :1338-1341
            }
        aggS[chunk][px] = a;
    }
    __syncthreads();
    // 2. the fold: sequential by nature (8 dependent multiplies per chunk); the aggregates are read 16 at a time so that the chain
    //    never waits for LDS
    if (tid < PX) {
        // (whole batches without a branch per chunk: this chain is what the kernel waits for)
        // The other waves emit chunk c as soon as carryS[c - 1] is published: the emissions run beside the fold instead of behind it
        // (-1.1 us).  LDS operations of one wave are carried out in order, so a reader that sees the counter sees the carries stored
        // before it.
        float c = aggS[0][tid];
        carryS[0][tid] = c;
        SGZ_PUBLISH(1);
        uint32_t d = 1;
        for (; d + 8 <= prm.numChunks; d += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = aggS[d + j][tid];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i) c = c * pole;        // every chunk before the last is full
                if (v[j] > c) c = v[j];
                carryS[d + j][tid] = c;
                if (j & 1) SGZ_PUBLISH(d + j + 1);
            }
        }
        for (; d < prm.numChunks; ++d) {
            const float v = aggS[d][tid];
#pragma unroll
            for (int i = 0; i < kMaxChunk; ++i) c = c * pole;
            if (v > c) c = v;
            carryS[d][tid] = c;
            SGZ_PUBLISH(d + 1);
        }
    }
    if (tid < 64) return;                                       // the fold's wave is done
    for (uint32_t e = tid - 64; e < items; e += 960) {
        const uint32_t px = e % PX, f = e / PX, chunk = f / kMaxChunk, t = f % kMaxChunk;
        const uint32_t pixel = pixelGroup() * PX + px;
        if (pixel >= prm.P) continue;
        awaitCarries(progressAddr, chunk);
        float a = (chunk == 0 && prm.stateIn) ? prm.stateIn[size_t(pixel) * 2] : 0.f;
        float cr = chunk > 0 ? carryS[chunk - 1][px] : 0.f;    // exact state at the end of the previous chunk
#pragma unroll
        for (int i = 0; i < kMaxChunk; ++i)
            if (uint32_t(i) <= t) {
                const float m = magS[chunk * kMaxChunk + i][px];
                a = a * pole;
                if (m > a) a = m;
                cr = cr * pole;
            }
        const float st = a > cr ? a : cr;
        float cb[3] = {0.f, 0.f, 0.f};                          // colourBuffer, SpectrumDSP.cpp:170-174
        blendColour(cb, dbMap(prm.slope[pixel], st, prm.sc, logTab), prm.colourTables, prm.sc);
        reinterpret_cast<uchar4 *>(prm.rgba)[size_t(f) * prm.P + pixel] = toRgba8(cb);
    }
}

// The same single launch for the step the reference performs on every frame -- line results of both graphs and the decay state after
// the last frame, with or without the colour column -- for one pair and at most kFusedChunks chunks.  A 1024-thread workgroup owns PX
// pixels for the whole time axis and all SIDES x G (side, graph) combinations of them: magnitudes of both sides and the carry-in state
// into LDS (late pixels of a channel-split K_A completed on the way, late_fix.hpp), zero-carry chunk scans, the exact sequential fold
// per combination, then the frames x PX emissions shared by all threads (replay, dB map of every combination, lines, colour of
// (side 0, graph 0), the new state at the last frame).  Replaces realLateKernel + decayLocalCarryKernel + decayEmitKernel.
template <int PX, int SIDES>
__global__ void __launch_bounds__(1024) decayFullFusedKernel(const DecayParams prm)
{
    constexpr int NCMB = SIDES * G;                             // combination m = side * G + graph
    __shared__ float aggS[kFusedChunks][NCMB][PX];
    __shared__ float carryS[kFusedChunks][NCMB][PX];
    __shared__ float magS[kFusedChunks * kMaxChunk][SIDES][PX];
    __shared__ float stIn[NCMB][PX];
    __shared__ uint32_t progress;                               // carryS[d] is final for every d < progress (all combinations)
    __shared__ double logTab[16][2];
    stageLogfTable(logTab);
    const uint32_t progressAddr = uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)&progress));
    if (threadIdx.x == 0) progress = 0;
    const uint32_t tid = threadIdx.x;
    const size_t perFrame = size_t(SIDES) * prm.P;              // (one pair)
    const uint32_t items = uint32_t(prm.frames) * PX;
    // 1a. magnitudes of both sides, carry-in state
    {
        constexpr int ROUNDS = kFusedChunks * kMaxChunk * PX * SIDES / 1024;      // (all loads requested before the first wait, as above)
        float m[ROUNDS], nyRe[ROUNDS], nyIm[ROUNDS], best[ROUNDS];
        const uint32_t lowestFix = prm.late.fixFrom0 < prm.late.fixFrom1 ? prm.late.fixFrom0 : prm.late.fixFrom1;
        const bool lateGroup = prm.hasLate && pixelGroup() * PX + PX > lowestFix;             // (uniform)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = tid + 1024u * r, px = e % PX, side = (e / PX) % SIDES, f = e / (PX * SIDES);
            const uint32_t pixel = pixelGroup() * PX + px;
            const bool in = f < uint32_t(prm.frames) && pixel < prm.P;
            m[r] = in ? prm.mapped[size_t(f) * perFrame + size_t(side) * prm.P + pixel] : 0.f;
            nyRe[r] = nyIm[r] = 0.f; best[r] = __builtin_inff();
            if (lateGroup && in && pixel >= (side ? prm.late.fixFrom1 : prm.late.fixFrom0)) {
                nyRe[r] = prm.late.ny[2 * size_t(f)]; nyIm[r] = prm.late.ny[2 * size_t(f) + 1];
                best[r] = lateBestSquare(prm.late, long(f), int(side), pixel);
            }
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint32_t e = tid + 1024u * r, px = e % PX, side = (e / PX) % SIDES, f = e / (PX * SIDES);
            const uint32_t pixel = pixelGroup() * PX + px;
            const bool in = f < uint32_t(prm.frames) && pixel < prm.P;
            float v = m[r] * prm.magScale;
            if (lateGroup && in && pixel >= (side ? prm.late.fixFrom1 : prm.late.fixFrom0))
                v = lateNyquistPixel(prm.late, lateNyquistValue(prm.late, nyRe[r], nyIm[r]), best[r], v);
            magS[f][side][px] = v;
        }
    }
    if (tid < NCMB * PX) {
        const uint32_t px = tid % PX, m = tid / PX, side = m / G, k = m % G;
        const uint32_t pixel = pixelGroup() * PX + px;
        stIn[m][px] = (prm.stateIn && pixel < prm.P) ? prm.stateIn[((size_t(k)) * prm.P + pixel) * 2 + side] : 0.f;
    }
    __syncthreads();
    // 1b. zero-carry scan of every (chunk, combination, pixel); chunk 0 starts from the carry-in
    if (tid < kFusedChunks * NCMB * PX) {
        const uint32_t px = tid % PX, m = (tid / PX) % NCMB, chunk = tid / (PX * NCMB);
        const uint32_t side = m / G, k = m % G;
        const float pole = prm.sc.pole[k];
        const long f0 = long(chunk) * kMaxChunk;
        const int len = chunk < prm.numChunks ? int(min(long(kMaxChunk), prm.frames - f0)) : 0;
        float a = chunk == 0 ? stIn[m][px] : 0.f;
#pragma unroll
        for (int t = 0; t < kMaxChunk; ++t)
            if (t < len) {
                const float v = magS[chunk * kMaxChunk + t][side][px];
                a = a * pole;                                   // states[i] *= pole, TransformDSP.inl:1336,:1370
                if (v > a) a = v;                               // :1338-1341
            }
        aggS[chunk][m][px] = a;
    }
    __syncthreads();
    // 2. the fold, one thread per (combination, pixel)
    if (tid < NCMB * PX) {
        const uint32_t px = tid % PX, m = tid / PX;
        const float pole = prm.sc.pole[m % G];
        float c = aggS[0][m][px];
        carryS[0][m][px] = c;
        SGZ_PUBLISH(1);
        uint32_t d = 1;
        for (; d + 8 <= prm.numChunks; d += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = aggS[d + j][m][px];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i) c = c * pole;      // every chunk before the last is full
                if (v[j] > c) c = v[j];
                carryS[d + j][m][px] = c;
                if (j & 1) SGZ_PUBLISH(d + j + 1);
            }
        }
        for (; d < prm.numChunks; ++d) {
            const float v = aggS[d][m][px];
#pragma unroll
            for (int i = 0; i < kMaxChunk; ++i) c = c * pole;
            if (v > c) c = v;
            carryS[d][m][px] = c;
            SGZ_PUBLISH(d + 1);
        }
        // the state after the last frame IS the fold's last value (the chunked form is the recurrence itself: multiplying by the pole
        // is monotone, so max(local scan, decayed carry) = the sequential state)
        // (the loop above decays by a full chunk in front of every aggregate: right for the carries the emissions read, not for a
        //  last chunk that is shorter)
        const uint32_t pixel = pixelGroup() * PX + px;
        if (prm.state && pixel < prm.P) {
            float fin = c;
            if (prm.numChunks > 1) {
                const int lenLast = int(prm.frames - long(prm.numChunks - 1) * kMaxChunk);
                fin = carryS[prm.numChunks - 2][m][px];
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i)
                    if (i < lenLast) fin = fin * pole;
                const float v = aggS[prm.numChunks - 1][m][px];
                if (v > fin) fin = v;
            }
            prm.state[(size_t(m % G) * prm.P + pixel) * 2 + m / G] = fin;
        }
    }
    if (tid < 64) return;                                       // (the fold's wave; the others emit beside it)
    // 3. emissions: one WALKER per (chunk, pixel, graph, half of the chunk).  It starts from the exact state at the end of the previous
    //    chunk and runs the reference's recurrence itself (state *= pole; if (mag > state) state = mag, TransformDSP.inl:1336-1341 --
    //    max(local scan, decayed carry) of the per-frame form is that recurrence, the decay being monotone), so a frame costs three
    //    operations per side before its dB map instead of a replay of the chunk up to it, and the carries are awaited once per walker.
    if (!prm.lines) {
        // no line results wanted (the image and the state after the last frame: what the reference holds when the same audio has gone
        // through): only (side 0, main graph) is emitted, one item per (frame, pixel) as in decayColourFusedKernel -- a walker's four
        // dB maps in a row would be the launch's critical path here
        if (!prm.rgba) return;
        const float pole = prm.sc.pole[0];
        for (uint32_t e = tid - 64; e < items; e += 960) {
            const uint32_t px = e % PX, f = e / PX, chunk = f / kMaxChunk, t = f % kMaxChunk;
            const uint32_t pixel = pixelGroup() * PX + px;
            if (pixel >= prm.P) continue;
            awaitCarries(progressAddr, chunk);
            float a = chunk == 0 ? stIn[0][px] : 0.f;
            float cr = chunk > 0 ? carryS[chunk - 1][0][px] : 0.f;   // exact state at the end of the previous chunk
#pragma unroll
            for (int i = 0; i < kMaxChunk; ++i)
                if (uint32_t(i) <= t) {
                    const float m = magS[chunk * kMaxChunk + i][0][px];
                    a = a * pole;
                    if (m > a) a = m;
                    cr = cr * pole;
                }
            const float st = a > cr ? a : cr;
            float cb[3] = {0.f, 0.f, 0.f};                      // colourBuffer, SpectrumDSP.cpp:170-174
            blendColour(cb, dbMap(prm.slope[pixel], st, prm.sc, logTab), prm.colourTables, prm.sc);
            reinterpret_cast<uchar4 *>(prm.rgba)[size_t(f) * prm.P + pixel] = toRgba8(cb);
        }
        return;
    }
    constexpr int HALF = kMaxChunk / 2;
    const uint32_t walkers = prm.numChunks * PX * G * 2;
    for (uint32_t w = tid - 64; w < walkers; w += 960) {
        // chunk-major: the first threads take the chunks whose carries are final first
        const uint32_t h = w & 1u, k = (w >> 1) % G, px = (w / (2 * G)) % PX, chunk = w / (2 * G * PX);
        const uint32_t pixel = pixelGroup() * PX + px;
        if (pixel >= prm.P) continue;
        awaitCarries(progressAddr, chunk);
        const float slope = prm.slope[pixel];
        const float pole = prm.sc.pole[k];
        const long f0 = long(chunk) * kMaxChunk;
        const int len = int(min(long(kMaxChunk), prm.frames - f0));
        float st[SIDES];
#pragma unroll
        for (int side = 0; side < SIDES; ++side) st[side] = chunk == 0 ? stIn[side * G + k][px] : carryS[chunk - 1][side * G + k][px];
        if (h) {                                                 // the second half passes over the first half's frames
#pragma unroll
            for (int i = 0; i < HALF; ++i)
                if (i < len) {
#pragma unroll
                    for (int side = 0; side < SIDES; ++side) {
                        const float v = magS[f0 + i][side][px];
                        st[side] = st[side] * pole;
                        if (v > st[side]) st[side] = v;
                    }
                }
        }
        const int i0 = h ? HALF : 0;
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int i = i0 + j;
            if (i >= len) break;
            const long f = f0 + i;
            float res[2] = {0.f, 0.f};                          // (results[i].phase = 0 in the one-channel modes, :1347)
#pragma unroll
            for (int side = 0; side < SIDES; ++side) {
                const float v = magS[f][side][px];
                st[side] = st[side] * pole;                     // states[i] *= pole, TransformDSP.inl:1336,:1370
                if (v > st[side]) st[side] = v;                 // :1338-1341
                if (prm.lines || (side == 0 && k == 0 && prm.rgba)) res[side] = dbMap(slope, st[side], prm.sc, logTab);
            }
            if (prm.lines) reinterpret_cast<float2 *>(prm.lines)[(size_t(f) * G + k) * prm.P + pixel] = float2{res[0], res[1]};
            if (prm.rgba && k == 0) {
                float cb[3] = {0.f, 0.f, 0.f};                  // colourBuffer, SpectrumDSP.cpp:170-174
                blendColour(cb, res[0], prm.colourTables, prm.sc);
                reinterpret_cast<uchar4 *>(prm.rgba)[size_t(f) * prm.P + pixel] = toRgba8(cb);
            }
        }
    }
}

// K_B2 (after K_B1 + K_B1b): one thread per (frame, pixel); a workgroup is one chunk x 32 pixels, so the GPU sees frames*P
// threads (the fp64 log of dbMap is ~200 instructions; with one thread per (chunk, pixel) a wave would issue eight of them
// back to back on an otherwise empty SIMD).  A state-only pass (no colour, no lines: the multi-GPU carry exchange) only
// needs the workgroups of the last chunk.
__global__ void __launch_bounds__(256) decayEmitKernel(const DecayParams prm, const uint32_t firstChunk)
{
    const int px = threadIdx.x & 31, t = threadIdx.x >> 5;
    const uint32_t groups = (prm.P + 31) / 32;
    const uint32_t chunk = firstChunk + blockIdx.x / groups;
    const uint32_t pixel = (blockIdx.x - (chunk - firstChunk) * groups) * 32 + px;
    const long f0 = long(chunk) * kMaxChunk;
    const long f1 = min(f0 + long(kMaxChunk), prm.frames);
    if (pixel >= prm.P || t >= int(f1 - f0)) return;
    const bool allCombos = prm.lines || (f1 == prm.frames && prm.state);
    const float *carryIn = chunk > 0 ? prm.agg + size_t(chunk - 1) * prm.C * prm.sides * G * prm.P + pixel : nullptr;
    emitPixel(prm, chunk, t, pixel, allCombos, carryIn, prm.P, prm.colourTables, prm.slope[pixel], nullptr);
}

// ---- SpectrumChannels::Phase (mapAndTransformDFTFilters, TransformDSP.inl:1393-1432) --------------------------------------
// mapped plane 0 = magnitude, plane 1 = cancellation.  state / lines hold (magnitude, phase) where the other modes hold
// (left, right).  The magnitude is the usual peak decay, the phase a one-pole smoother  s = p + pole^0.3 (s - p)  of
// p = cancellation * mag (multiplied again for every graph: quirk Q7) -- a linear fp32 recurrence whose rounding depends on
// the order, so this kernel walks the frames sequentially: one thread per (pair, pixel), magnitudes fetched 8 frames ahead.
// Only the recurrences run here (C * P threads for the whole time axis): the filter states go out raw, and the fp64 log of
// the dB map is applied by kernels with one thread per value (decayPhaseLinesKernel, decayPhaseColourKernel).
__global__ void __launch_bounds__(256) decayPhaseScanKernel(const DecayParams prm, float *work /*[frames][C][P] main-graph state*/)
{
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(prm.C) * prm.P) return;
    const uint32_t pair = uint32_t(gid / prm.P), pixel = uint32_t(gid - size_t(pair) * prm.P);
    const size_t perFrame = size_t(prm.C) * 2 * prm.P;
    const float *src = prm.mapped + size_t(pair) * 2 * prm.P + pixel;
    float sm[G], sp[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        sm[k] = prm.stateIn ? prm.stateIn[((size_t(pair) * G + k) * prm.P + pixel) * 2 + 0] : 0.f;
        sp[k] = prm.stateIn ? prm.stateIn[((size_t(pair) * G + k) * prm.P + pixel) * 2 + 1] : 0.f;
    }
    // batches of kMaxChunk frames; the next batch's loads are in flight while this one runs through the recurrences
    float m[kMaxChunk], c[kMaxChunk], mn[kMaxChunk], cn[kMaxChunk];
    auto fetch = [&](long f0, float (&mm)[kMaxChunk], float (&cc)[kMaxChunk]) {
#pragma unroll
        for (int i = 0; i < kMaxChunk; ++i) {
            const long f = f0 + i < prm.frames ? f0 + i : prm.frames - 1;
            mm[i] = src[size_t(f) * perFrame];
            cc[i] = src[size_t(f) * perFrame + prm.P];
        }
    };
    fetch(0, m, c);
    // the launch has C * P threads (a lone wave per SIMD issues one instruction per ~5 clocks): the loop is written for
    // instruction count -- output pointers bumped per frame, the graph constants in registers
    static_assert(G == 2, "the two graphs are written out explicitly below");
    float *wp = work + size_t(pair) * prm.P + pixel;
    const size_t wStride = size_t(prm.C) * prm.P;
    float *lp = prm.lines ? prm.lines + (size_t(pair) * G * prm.P + pixel) * 2 : nullptr;
    const size_t lStride = size_t(prm.C) * G * prm.P * 2, lGraph = size_t(prm.P) * 2;
    const float pole0 = prm.sc.pole[0], pole1 = prm.sc.pole[1], pp0 = prm.sc.phasePole[0], pp1 = prm.sc.phasePole[1];
    for (long f0 = 0; f0 < prm.frames; f0 += kMaxChunk) {
        const bool more = f0 + kMaxChunk < prm.frames;
        if (more) fetch(f0 + kMaxChunk, mn, cn);
        const int n = int(prm.frames - f0 < long(kMaxChunk) ? prm.frames - f0 : long(kMaxChunk));
#pragma unroll
        for (int i = 0; i < kMaxChunk; ++i) {
            if (i >= n) break;
            const float mag = m[i] * 0.5f;                          // mag *= consts::half, :1407
            float phase = c[i];
            sm[0] = sm[0] * pole0;
            if (mag > sm[0]) sm[0] = mag;
            phase = phase * mag;                                    // inside the graph loop (Q7)
            sp[0] = phase + pp0 * (sp[0] - phase);
            sm[1] = sm[1] * pole1;
            if (mag > sm[1]) sm[1] = mag;
            phase = phase * mag;
            sp[1] = phase + pp1 * (sp[1] - phase);
            if (lp) {
                lp[0] = sm[0]; lp[1] = sp[0];
                lp[lGraph] = sm[1]; lp[lGraph + 1] = sp[1];
                lp += lStride;
            }
            *wp = sm[0];
            wp += wStride;
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < kMaxChunk; ++i) { m[i] = mn[i]; c[i] = cn[i]; }
        }
    }
    if (prm.state) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            prm.state[((size_t(pair) * G + k) * prm.P + pixel) * 2 + 0] = sm[k];
            prm.state[((size_t(pair) * G + k) * prm.P + pixel) * 2 + 1] = sp[k];
        }
    }
}

// lines: raw (magnitude, phase) filter states -> dB, in place; one thread per value
__global__ void __launch_bounds__(256) decayPhaseLinesKernel(const DecayParams prm)
{
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(prm.frames) * prm.C * G * prm.P * 2) return;
    const uint32_t pixel = uint32_t((gid >> 1) % prm.P);
    prm.lines[gid] = dbMap(prm.slope[pixel], prm.lines[gid], prm.sc);
}

// colour columns from the main graph's magnitude states: one thread per (frame, pixel), dB map, pairs blended in order
__global__ void __launch_bounds__(256) decayPhaseColourKernel(const DecayParams prm, const float *work)
{
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(prm.frames) * prm.P) return;
    const long f = long(gid / prm.P);
    const uint32_t pixel = uint32_t(gid - size_t(f) * prm.P);
    float cb[3] = {0.f, 0.f, 0.f};
    const float slope = prm.slope[pixel];
    for (uint32_t pair = 0; pair < prm.C; ++pair)
        blendColour(cb, dbMap(slope, work[(size_t(f) * prm.C + pair) * prm.P + pixel], prm.sc), prm.colourTables + size_t(pair) * NC * 3,
                    prm.sc);
    reinterpret_cast<uchar4 *>(prm.rgba)[gid] = toRgba8(cb);
}

hipError_t launchDecayPhase(const DecayParams &prm, float *work, hipStream_t stream)
{
    const size_t n1 = size_t(prm.C) * prm.P;
    hipLaunchKernelGGL(decayPhaseScanKernel, dim3(unsigned((n1 + 255) / 256)), dim3(256), 0, stream, prm, work);
    if (prm.lines) {
        const size_t n3 = size_t(prm.frames) * prm.C * G * prm.P * 2;
        hipLaunchKernelGGL(decayPhaseLinesKernel, dim3(unsigned((n3 + 255) / 256)), dim3(256), 0, stream, prm);
    }
    if (prm.rgba) {
        const size_t n2 = size_t(prm.frames) * prm.P;
        hipLaunchKernelGGL(decayPhaseColourKernel, dim3(unsigned((n2 + 255) / 256)), dim3(256), 0, stream, prm, work);
    }
    return hipGetLastError();
}

// carry-in state of rank `rank` from every rank's zero-carry end state (see sgz.h, sgz_decay_fold_carry)
struct FoldFrames { long long f[64]; };
__global__ void __launch_bounds__(256)
decayFoldKernel(const float *aggs, FoldFrames frames, uint32_t rank, size_t perRank, uint32_t P, float pole0, float pole1,
                float *carry)
{
    const size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= perRank) return;
    const uint32_t k = uint32_t((e / (size_t(P) * 2)) % G);
    const float pole = k == 0 ? pole0 : pole1;
    float c = 0.f;
    for (uint32_t q = 0; q < rank; ++q) {
        for (long long i = 0; i < frames.f[q]; ++i) c = c * pole;
        const float a = aggs[size_t(q) * perRank + e];
        if (a > c) c = a;
    }
    carry[e] = c;
}

hipError_t launchDecayFold(const float *aggs, const long long *framesPerRank, uint32_t world, uint32_t rank, size_t perRank,
                           uint32_t P, const DeviceScalars &sc, float *carry, hipStream_t stream)
{
    if (world > 64) return hipErrorInvalidValue;
    FoldFrames fr{};
    for (uint32_t q = 0; q < world; ++q) fr.f[q] = framesPerRank[q];
    const int block = 256;
    const unsigned grid = unsigned((perRank + block - 1) / block);
    hipLaunchKernelGGL(decayFoldKernel, dim3(grid), dim3(block), 0, stream, aggs, fr, rank, perRank, P, sc.pole[0], sc.pole[1], carry);
    return hipGetLastError();
}

// test hook (sgz_stage_logf): std::log(float) exactly as dbMap evaluates it, over an array
__global__ void __launch_bounds__(256) logfKernel(const float *x, float *y, size_t n)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) y[i] = glibcLogf(x[i]);
}
hipError_t launchLogf(const float *x, float *y, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(logfKernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, x, y, n);
    return hipGetLastError();
}

// Carry-apply pass of the multi-GPU exchange: the chunk aggregates were scanned from a ZERO carry-in; the true carry-in c0 of this
// rank arrives later.  By the monotonicity identity the state at the end of chunk d is max(agg[d], decay^{frames so far}(c0)), the
// decay done with the reference's sequential fp32 multiplies -- one thread per (pair, side, graph, pixel), no second scan of the
// mapped magnitudes.  carry: [C][G][P][2].
__global__ void __launch_bounds__(256) decayApplyCarryKernel(const DecayParams prm, const float *carry)
{
    const size_t per = size_t(prm.C) * prm.sides * G * prm.P;
    const size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= per) return;
    const uint32_t pixel = uint32_t(e % prm.P);
    const uint32_t k = uint32_t((e / prm.P) % G);
    const uint32_t ps = uint32_t(e / (size_t(prm.P) * G));
    const uint32_t pair = ps / prm.sides, side = ps - pair * prm.sides;
    const float pole = prm.sc.pole[k];
    float c = carry[((size_t(pair) * G + k) * prm.P + pixel) * 2 + side];
    for (uint32_t d = 0; d < prm.numChunks; ++d) {
        const long f0 = long(d) * kMaxChunk;
        const int len = int(min(long(kMaxChunk), prm.frames - f0));
        for (int i = 0; i < len; ++i) c = c * pole;
        float *a = prm.agg + size_t(d) * per + e;
        if (c > *a) *a = c;
    }
}
hipError_t launchDecayApplyCarry(const DecayParams &prm, const float *carry, hipStream_t stream)
{
    const size_t per = size_t(prm.C) * prm.sides * G * prm.P;
    hipLaunchKernelGGL(decayApplyCarryKernel, dim3(unsigned((per + 255) / 256)), dim3(256), 0, stream, prm, carry);
    return hipGetLastError();
}

hipError_t launchDecayLocalCarry(const DecayParams &prm, hipStream_t stream)
{
    if (prm.numChunks > uint32_t(kFusedChunks)) {                       // (long renders scan everything: colourOnly is the fused kernel's)
        hipError_t e = launchDecayLocal(prm, stream);
        return e != hipSuccess ? e : launchDecayCarry(prm, stream);
    }
    const size_t entries = prm.colourOnly ? size_t(prm.C) * prm.P : size_t(prm.C) * prm.sides * prm.P;
    auto launch = [&](auto ch) {
        constexpr int CH = decltype(ch)::value;
        hipLaunchKernelGGL(decayLocalCarryKernel<CH>, dim3(unsigned((entries + 1024 / CH - 1) / (1024 / CH))), dim3(1024), 0, stream, prm);
    };
    if (prm.numChunks <= 8) launch(std::integral_constant<int, 8>{});
    else if (prm.numChunks <= 16) launch(std::integral_constant<int, 16>{});
    else if (prm.numChunks <= 32) launch(std::integral_constant<int, 32>{});
    else launch(std::integral_constant<int, 64>{});
    return hipGetLastError();
}

hipError_t launchDecayLocal(const DecayParams &prm, hipStream_t stream)
{
    const size_t total = size_t(prm.C) * prm.sides * prm.P * prm.numChunks;
    const int block = 256;
    const unsigned grid = unsigned((total + block - 1) / block);
    hipLaunchKernelGGL(decayLocalKernel, dim3(grid), dim3(block), 0, stream, prm);
    return hipGetLastError();
}

hipError_t launchDecayCarry(const DecayParams &prm, hipStream_t stream)
{
    const size_t total = size_t(prm.C) * prm.sides * G * prm.P;
    const int block = 256;
    const unsigned grid = unsigned((total + block - 1) / block);
    if (total * prm.numChunks * sizeof(float) < (size_t(1) << 32))
        hipLaunchKernelGGL(decayCarryKernel<uint32_t>, dim3(grid), dim3(block), 0, stream, prm);
    else
        hipLaunchKernelGGL(decayCarryKernel<size_t>, dim3(grid), dim3(block), 0, stream, prm);
    return hipGetLastError();
}

bool decayColourFusedApplies(const DecayParams &prm)
{
    return prm.rgba && !prm.lines && !prm.state && prm.C == 1 && prm.numChunks <= uint32_t(kFusedChunks);
}
bool decayFullFusedApplies(const DecayParams &prm)
{
    return (prm.lines || prm.state) && prm.C == 1 && prm.numChunks <= uint32_t(kFusedChunks) && (prm.sides == 1 || prm.sides == 2);
}
hipError_t launchDecayFullFused(const DecayParams &prm, hipStream_t stream)
{
    constexpr int PX = 4;
    if (prm.sides == 2) hipLaunchKernelGGL((decayFullFusedKernel<PX, 2>), dim3((prm.P + PX - 1) / PX), dim3(1024), 0, stream, prm);
    else hipLaunchKernelGGL((decayFullFusedKernel<PX, 1>), dim3((prm.P + PX - 1) / PX), dim3(1024), 0, stream, prm);
    return hipGetLastError();
}
hipError_t launchDecayColourFused(const DecayParams &prm, hipStream_t stream)
{
    if (prm.fusedPixels == 16) hipLaunchKernelGGL(decayColourFusedKernel<16>, dim3((prm.P + 15) / 16), dim3(1024), 0, stream, prm);
    else if (prm.fusedPixels == 8) hipLaunchKernelGGL(decayColourFusedKernel<8>, dim3((prm.P + 7) / 8), dim3(1024), 0, stream, prm);
    else hipLaunchKernelGGL(decayColourFusedKernel<4>, dim3((prm.P + 3) / 4), dim3(1024), 0, stream, prm);
    return hipGetLastError();
}

// Several pairs: the same outputs with the pairs side by side.  decayEmitKernel's thread owns one (frame, pixel) and walks the pairs one
// after the other -- 32 dependent rounds of (loads, replay of the chunk up to its frame, dB map, blend) at cfg5, 114 us for 91 MB.  Here a
// thread owns (pair, pixel) for the 8 frames of the chunk: it reads each magnitude once, runs the recurrence once from the exact
// carry (max(a, decayed carry) of emitPixel is that recurrence: the decay is monotone, D(max(x, y)) = max(D(x), D(y))), and leaves
// every frame's colour contribution in LDS; the (frame, pixel) owners then blend 8 pairs' contributions in pair order (the screen
// blend is the only step that couples the pairs, SpectrumDSP.cpp:170-174; a pair with nothing to add contributes zeros:
// cb + (1 - cb) 0 == cb).
__global__ void __launch_bounds__(256) decayEmitPairsKernel(const DecayParams prm, const uint32_t firstChunk)
{
    constexpr int PL = 8;                                       // pairs per pass
    __shared__ float cv[kMaxChunk][PL][32][3];
    __shared__ float st[kMaxChunk][256];
    __shared__ double logTab[16][2];
    stageLogfTable(logTab);
    static_assert(G == 2, "two line graphs");
    const int px = threadIdx.x & 31, pl = threadIdx.x >> 5;     // pl: pair lane (contributions) / frame in the chunk (blend)
    const uint32_t groups = (prm.P + 31) / 32;
    const uint32_t chunk = firstChunk + blockIdx.x / groups;
    const uint32_t pixel = (blockIdx.x - (chunk - firstChunk) * groups) * 32 + px;
    const long f0 = long(chunk) * kMaxChunk;
    const int len = int(min(long(kMaxChunk), prm.frames - f0));
    const bool live = pixel < prm.P;
    const bool allCombos = prm.lines || (f0 + len == prm.frames && prm.state);
    const size_t perFrame = size_t(prm.C) * prm.sides * prm.P;
    const float slope = live ? prm.slope[pixel] : 0.f;
    float cb[3] = {0.f, 0.f, 0.f};                              // colourBuffer of (frame f0 + pl, pixel)
    for (uint32_t p0 = 0; p0 < prm.C; p0 += PL) {
        const uint32_t pair = p0 + pl;
#pragma unroll
        for (int i = 0; i < kMaxChunk; ++i) { cv[i][pl][px][0] = 0.f; cv[i][pl][px][1] = 0.f; cv[i][pl][px][2] = 0.f; }
        if (live && pair < prm.C) {
            const float *sca = prm.colourTables + size_t(pair) * NC * 3;
#pragma unroll 1
            for (uint32_t side = 0; side < prm.sides; ++side) {
                if (!allCombos && side != 0) continue;          // only (side 0, graph 0) feeds the colour column
                const uint32_t ps = pair * prm.sides + side;
                float mag[kMaxChunk];
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i)
                    mag[i] = i < len ? prm.mapped[size_t(f0 + i) * perFrame + size_t(ps) * prm.P + pixel] * prm.magScale : 0.f;
#pragma unroll 1
                for (int k = 0; k < G; ++k) {
                    const bool colour = (side == 0 && k == 0 && prm.rgba);
                    if (!colour && !allCombos) continue;
                    const float pole = k ? prm.sc.pole[G - 1] : prm.sc.pole[0];
                    float s = chunk > 0 ? prm.agg[(size_t(chunk - 1) * prm.C * prm.sides * G + (ps * G + k)) * prm.P + pixel]
                                        : (prm.stateIn ? prm.stateIn[((size_t(pair) * G + k) * prm.P + pixel) * 2 + side] : 0.f);
                    // the chunk's states (unrolled: the chain), parked in LDS so that the per-frame outputs below are ONE copy of the
                    // dB map's code in a rolled loop (unrolled over sides x graphs x frames the kernel needed 136 registers)
#pragma unroll
                    for (int i = 0; i < kMaxChunk; ++i) {
                        s = s * pole;                           // states[i] *= pole, TransformDSP.inl:1336,:1370
                        if (mag[i] > s) s = mag[i];             // :1338-1341
                        st[i][threadIdx.x] = s;
                    }
#pragma unroll 1
                    for (int i = 0; i < len; ++i) {
                        const float si = st[i][threadIdx.x];
                        const long f = f0 + i;
                        if (prm.state && f == prm.frames - 1) prm.state[((size_t(pair) * G + k) * prm.P + pixel) * 2 + side] = si;
                        if (!colour && !prm.lines) continue;
                        const float result = dbMap(slope, si, prm.sc, logTab);
                        if (prm.lines) {
                            float *lr = prm.lines + (((size_t(f) * prm.C + pair) * G + k) * prm.P + pixel) * 2;
                            lr[side] = result;
                            if (prm.sides == 1) lr[1] = 0.f;    // results[i].phase = 0 in the one-channel modes (:1347)
                        }
                        float c3[3];
                        if (colour && colourOf(result, sca, prm.sc, c3)) { cv[i][pl][px][0] = c3[0]; cv[i][pl][px][1] = c3[1]; cv[i][pl][px][2] = c3[2]; }
                    }
                }
            }
        }
        __syncthreads();
        if (prm.rgba) {
#pragma unroll
            for (int q = 0; q < PL; ++q)
                if (p0 + q < prm.C) { const float c3[3] = {cv[pl][q][px][0], cv[pl][q][px][1], cv[pl][q][px][2]}; screenBlend(cb, c3); }
        }
        __syncthreads();
    }
    if (prm.rgba && live && pl < len) reinterpret_cast<uchar4 *>(prm.rgba)[size_t(f0 + pl) * prm.P + pixel] = toRgba8(cb);
}

hipError_t launchDecayEmit(const DecayParams &prm, hipStream_t stream)
{
    const uint32_t firstChunk = (prm.rgba || prm.lines) ? 0u : prm.numChunks - 1;        // state-only pass: last chunk
    const unsigned grid = unsigned(size_t((prm.P + 31) / 32) * (prm.numChunks - firstChunk));
    static_assert(kMaxChunk == 8, "decayEmitPairsKernel: 8 frames x 32 pixels = one workgroup");
    if (prm.C >= 2) hipLaunchKernelGGL(decayEmitPairsKernel, dim3(grid), dim3(256), 0, stream, prm, firstChunk);
    else hipLaunchKernelGGL(decayEmitKernel, dim3(grid), dim3(256), 0, stream, prm, firstChunk);
    return hipGetLastError();
}

}  // namespace sgz
