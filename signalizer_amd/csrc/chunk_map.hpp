// chunk_map.hpp -- the pixel mapping of mapToLinearSpace (TransformDSP.inl:565-639, :871-985) for the channel-split kernels
// (spectrum_real.hip), on one side's magnitudes S[0 .. M] held in LDS at chunkPos(i) (plan.hpp).  gfx950 only.
//
// What the mapping is: ~40 % of the pixels of a log view interpolate <= 10 neighbouring bins (Lanczos / linear / none); every other
// pixel takes the entry with the largest square over a run of bins ("first strictly greater |X|^2", :957-979), and consecutive
// pixels' runs tile the spectrum: M values go in, a few hundred maxima come out.  fl(m^2) is strictly increasing in |m| while the
// square is a normal float, so the value of such a pixel is simply max |S| over its run (stft_body.hpp has the argument and the
// literal replay for runs below 2^-62, kept here unchanged).
//
// How it is computed: thread t owns the chunk S[32 t .. 32 t + 31] (16 ds_read_b64) and runs a SEGMENTED running maximum over it in
// registers.  The segments are the plan's tiles (runs of >= 2 entries); where they end is the same for every frame, so the plan
// ships, per thread, the 32 bits "element j of this chunk ends a tile" (chunkEnds); a vector compare turns bit j into the wave's
// lane mask in an SGPR pair.  Element j then costs
//     exec = ~ends[j - 1];  v[j] = max(v[j], v[j - 1])          (lanes whose previous element closed a tile start over)
// and where some lane closes a tile (ends[j] != 0, a scalar test) those lanes store the running value to the tile's slot:
//     exec = ends[j];  RE[slot++] = v[j]
// The last element's running value goes to CE[t].  A pixel's maximum is then max(RE[its tile], CE[c] for the chunks c its run
// covers before the last one): one thread per pixel, a handful of LDS reads.  No per-element masks, no per-piece work list, no
// atomics; S is never modified, so the literal replay still finds the original entries.
// (The round-2 map cut the runs into 16-bin pieces with per-element bit-field masks and a three-stage resolve: 9 k of a
// workgroup's 34.6 k clocks, tools/phase_clocks.py.)
#pragma once
#include "stft_body.hpp"

namespace sgz {

struct ChunkTables {
    const uint32_t *ends;       // [T] of this side: bit j = element j of the thread's chunk closes a tile
    const uint32_t *reBase;     // [T]
    const uint2 *crec;          // [P]
    const PixelRec *recs;       // [P]
    const float *weights12;     // [][kTapFloats]
    float *out;                 // [P] or null
    float *bestOut;             // [64] or null: the winning SQUARE of the pixels >= bestFrom goes here as well (realLateKernel compares it with csf[N/2]'s)
    int bestFrom;
    int P;
    bool right;                 // the side holds csf[M .. N] (scan order of the literal replay: descending index)
    unsigned long long *clk;    // -DSGZ_DEBUG: this wave's row of the phase clocks (slots 10 .. 12), or null
};
#ifdef SGZ_DEBUG
#define SGZ_MAPCLK(slot) do { if (tb.clk && (tid & 63) == 0) tb.clk[16 * (tid >> 6) + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SGZ_MAPCLK(slot) do { } while (0)
#endif

// One instance per thread.  prefetch(): the table reads (records, weights, slot base) -- issued before the barrier that completes S,
// so that the map starts with its operands in registers.  run(): lds = S at chunkPos(i); re / ce: the tile and chunk maxima (LDS,
// `slots + 1` and T floats).  Every thread of the workgroup calls both (full waves: the exec games below restore exec to all
// ones).  at(k): float position of csf[k] (literal replay only).
template <int T>
struct ChunkMap {
    static constexpr int RB = 2;                                        // pixels per thread in the first round (records stay in registers across the barrier)
    // Everything the map needs per pixel is in the two words of its ChunkRec (plan.hpp); the PixelRec -- 16 more bytes per pixel and
    // workgroup through the CU's fetch path -- is only read by the literal replay of a run without a normal square.
    uint2 cr[RB];
    uint32_t slotBase, endBits;

    __device__ __forceinline__ void prefetch(const ChunkTables &tb, int tid)
    {
        slotBase = tb.reBase[tid];
        endBits = tb.ends[tid];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int x = tid + b * T;
            cr[b] = x < tb.P ? tb.crec[x] : uint2{0u, kChunkOff};
        }
    }
    // The prefetched words are waited for HERE (a workgroup that walks over units requests the next unit's samples behind this point: a
    // later wait for these words would be a wait for those samples as well -- the counter of outstanding loads is in order)
    __device__ __forceinline__ void arrived()
    {
        asm volatile("" : "+v"(slotBase), "+v"(endBits));
#pragma unroll
        for (int b = 0; b < RB; ++b) asm volatile("" : "+v"(cr[b].x), "+v"(cr[b].y));
    }
    static __device__ __forceinline__ bool interpolated(const uint2 c) { return (c.y & kChunkInterp) != 0u; }

    // <= 10 taps as kTapFloats contiguous floats, accumulated in tap order; entries that are not taps carry weight +0 and read a finite
    // value -- a zeroed pad slot or a neighbouring bin -- so they add +-0 to a sum that is never -0: the same sum bit for bit
    static __device__ __forceinline__ float taps(const float *lds, uint32_t pos, const float4 (&q)[3])
    {
#pragma clang fp contract(off)
        const float w[kTapFloats] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w};
        const float *src = lds + pos;
        float m[kTapFloats];
#pragma unroll
        for (int i = 0; i < kTapFloats; ++i) m[i] = src[i];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < kTapFloats; ++i) { const float prod = m[i] * w[i]; acc = acc + prod; }
        return acc;
    }

    template <typename Index>
    static __device__ __forceinline__ void resolve(const ChunkTables &tb, const Index at, const float *lds, const float *re, const float *ce,
                                                   float invSize, const uint2 c, int x)
    {
#pragma clang fp contract(off)
        if (c.y & (kChunkInterp | kChunkOff)) return;
        float best = 0.f;
        if (c.y & kChunkDirect) best = __builtin_fabsf(lds[c.x]);
        else if (!(c.y & kChunkNoScan)) {
            best = re[c.x & 0xFFFFu];
            const int c0 = int(c.x >> 16), nC = int(c.y & 0xFFFFu);
#ifndef SGZ_RESOLVE_CHAIN
            // the run's tile maxima four at a time as independent requests (a pixel of the log view's top covers 4-8 tiles, and as a
            // dependent chain -- read, wait, max, read -- this was the longest thing behind the map's barrier: the last wave of a
            // workgroup left 1.5 k clocks after the first).  max is exact: any order gives the same value.  (Eight at a time spills a
            // register of the N = 65536 kernel.)
            // The first four go out together with the run's own partial maximum (re) -- an index past the run repeats its last tile, a run
            // without whole tiles (nC = 0) takes re in their place.
            {
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = ce[c0 + min(i, max(nC - 1, 0))];
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = nC > 0 ? e[i] : best;
                best = __builtin_fmaxf(__builtin_fmaxf(best, e[0]), __builtin_fmaxf(__builtin_fmaxf(e[1], e[2]), e[3]));
            }
            for (int i0 = 4; i0 < nC; i0 += 4) {
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = ce[c0 + min(i0 + i, nC - 1)];
                best = __builtin_fmaxf(__builtin_fmaxf(best, e[0]), __builtin_fmaxf(__builtin_fmaxf(e[1], e[2]), e[3]));
            }
#else
            for (int i = 0; i < nC; ++i) best = __builtin_fmaxf(best, ce[c0 + i]);
#endif
        }
        if (c.y & kChunkPlusM) best = __builtin_fmaxf(best, __builtin_fabsf(lds[at(tb.right ? at.size() : at.size() / 2)]));
        float val = best, bestSq = best * best + 0.f;                    // Math::square(csf[offset]) of the winner (imag == 0)
        if (!(best >= 0x1p-62f)) {
            // the run's squares are denormal, zero or NaN: distinct values can tie there -- the reference's scan, literally
            // (first strictly greater square in offset order; initial arg = bin, :953)
            const PixelRec r = tb.recs[x];
            const int N = at.size();
            int arg = r.c;
            bestSq = 0.f;
            for (int o = r.a; o < r.a + r.b; ++o) {
                const int k = tb.right ? N - o : o;
                const float mm = lds[at(k)];
                const float sq = mm * mm + 0.f;
                if (sq > bestSq) { bestSq = sq; arg = k; }
            }
            // (a right-side run without a positive square keeps arg = bin, a LEFT-side index (:953): this workgroup holds one side and
            // shows 0 -- what the run's own bins say; see stft_body.hpp)
            val = at.holds(arg) ? lds[at(arg)] : 0.f;
        }
        const float pix = finishPixel<5>(invSize * val);
        if (tb.bestOut && x >= tb.bestFrom) tb.bestOut[x - tb.bestFrom] = bestSq;
        if (tb.out) tb.out[x] = pix;
    }

    template <typename Index>
    __device__ __forceinline__ void run(const ChunkTables &tb, const Index at, const float *lds, float *re, float *ce, float invSize, int tid)
    {
        // ---- tap weights of the first round's pixels (unconditional, independent loads -- an arg-max pixel reads row 0; they are
        // consumed behind the scan, which hides their latency)
        // (a wave none of whose lanes interpolates -- the upper pixels of a log view -- neither fetches weights nor runs the taps)
        float4 wq[RB][3];
        bool anyInterp[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            anyInterp[b] = __builtin_amdgcn_ballot_w64(interpolated(cr[b])) != 0ull;
            if (anyInterp[b]) {
                const float4 *wp = reinterpret_cast<const float4 *>(tb.weights12 + size_t(interpolated(cr[b]) ? cr[b].x : 0u) * kTapFloats);
                wq[b][0] = wp[0]; wq[b][1] = wp[1]; wq[b][2] = wp[2];
            }
        }
        // ---- this thread's chunk
        float v[32];
        {
            v2 t[16];                                                   // sixteen single ds_read_b64 (fft_common.hpp ldsRead64: paired reads run at half the rate)
            ldsReadRun64<16, 8>(t, ldsAddress(lds + chunkPos(32 * tid)));
            ldsReadsDone(t);
#pragma unroll
            for (int j = 0; j < 16; ++j) { v[2 * j] = t[j].x; v[2 * j + 1] = t[j].y; }
        }
        v[0] = __builtin_fabsf(v[0]);                                   // csf[0] (left side, chunk 0) is a signed real; every other entry is a magnitude
        // ---- segmented running maximum + tile maxima.  The lane mask "element j closes a tile" is a compare of the thread's end bits
        // (two vector operations per element; fetching the masks ready-made with scalar loads cost four dependent memory round trips
        // per wave, ~2 k clocks of a workgroup's critical path)
        uint32_t slotAddr = (slotBase << 2) + uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)re));   // LDS byte address of the chunk's first tile slot
        uint64_t prev = 0;
        // (round 6: the mask of element j as the CARRY of a left shift -- v_add_co x, x of the bit-reversed word delivers the lane mask and
        // the shifted word in one vector operation, where "and with 1 << j, compare with 0" took two)
        uint32_t rev = __builtin_bitreverse32(endBits);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
#ifndef SGZ_SCAN_MASK_BY_COMPARE
            uint64_t e;
            asm volatile("v_add_co_u32_e64 %0, %1, %0, %0" : "+v"(rev), "=s"(e));
#else
            const uint64_t e = __builtin_amdgcn_uicmp(endBits & (1u << j), 0u, 33 /* ICMP_NE */);
#endif
            if (j > 0)
                asm volatile("s_andn2_b64 exec, exec, %2\n\tv_max_f32 %0, %0, %1\n\ts_mov_b64 exec, -1"
                             : "+v"(v[j]) : "v"(v[j - 1]), "s"(prev) : "scc");      // (s_andn2 writes SCC)
            if (e != 0)                                                 // wave-uniform
                asm volatile("s_mov_b64 exec, %2\n\tds_write_b32 %0, %1\n\tv_add_u32 %0, 4, %0\n\ts_mov_b64 exec, -1"
                             : "+v"(slotAddr) : "v"(v[j]), "s"(e) : "memory");
            prev = e;
        }
        ce[tid] = v[31];
        SGZ_MAPCLK(10);
        // ---- interpolated pixels of the first round
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            if (anyInterp[b]) {
                const float acc = taps(lds, interpolated(cr[b]) ? (cr[b].y & 0xFFFFu) : 0u, wq[b]);
                if (interpolated(cr[b]) && tb.out) tb.out[tid + b * T] = finishPixel<5>(invSize * acc);
            }
        }
        SGZ_MAPCLK(11);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the tile stores above are invisible to the compiler's counters
        ldsBarrier();
        SGZ_MAPCLK(12);
        // ---- one thread per arg-max pixel
#pragma unroll
        for (int b = 0; b < RB; ++b) resolve(tb, at, lds, re, ce, invSize, cr[b], tid + b * T);
        // ---- further rounds (more than RB T pixels per side)
        for (int base = RB * T; base < tb.P; base += T) {
            const int x = base + tid;
            if (x < tb.P) {
                const uint2 c = tb.crec[x];
                if (interpolated(c)) {
                    const float4 *wp = reinterpret_cast<const float4 *>(tb.weights12 + size_t(c.x) * kTapFloats);
                    const float4 q[3] = {wp[0], wp[1], wp[2]};
                    const float acc = taps(lds, c.y & 0xFFFFu, q);
                    if (tb.out) tb.out[x] = finishPixel<5>(invSize * acc);
                }
                resolve(tb, at, lds, re, ce, invSize, c, x);
            }
        }
    }
};

}  // namespace sgz
