// stft_body.hpp -- the body of K_A (window x audio -> FFT -> split -> |X| -> pixel mapping for one (frame, pair)); the kernel
// wrappers and the design notes are in spectrum_fft.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "complex_dc.hpp"
#include "fft_common.hpp"
#include "fft_scalar.hpp"
#include "kernels.hpp"
#include "late_fix.hpp"

namespace sgz {

// Debug plumbing (tools/phase_clocks.py, tools/ablate.py) exists only in a -DSGZ_DEBUG build (SGZ_EXTRA_HIPCC_FLAGS=-DSGZ_DEBUG python
// signalizer_amd/build.py --force): every wave of one workgroup stores the shader clock at the phase boundaries,
// phaseClock[16 * wave + slot], and prm.ablate switches phases off.  The shipped kernel carries none of it: no branch around a
// phase, nothing that keeps the scheduler from interleaving one phase's LDS traffic with the next phase's arithmetic.
#ifdef SGZ_DEBUG
#define SGZ_CLK(slot)                                                                                   \
    do {                                                                                                \
        if (prm.phaseClock && (tid & 63) == 0 && task == long(prm.ablate >> 16) && sgzClkHalf)          \
            prm.phaseClock[16 * (tid >> 6) + (slot)] = __builtin_readcyclecounter();                     \
    } while (0)
#define SGZ_ABLATED(bits) ((prm.ablate & (bits)) != 0u)
#define SGZ_CLK_HALF(expr) const bool sgzClkHalf = (expr)
#else
#define SGZ_CLK(slot) do { } while (0)
#define SGZ_ABLATED(bits) false
#define SGZ_CLK_HALF(expr) do { } while (0)
#endif
#define SGZ_WCLK(i) do { } while (0)

// Pixel mapping of mapToLinearSpace (TransformDSP.inl:565-639, :871-985) on the csf magnitudes held in LDS
// (bank-padded natural order).  Every operation rounds exactly like the reference's scalar fp32 code:
// contraction is off in these functions (NB: hip's __fmul_rn/__fadd_rn are plain * and + and would be fused,
// and __fsqrt_rn is the approximate native sqrt -- neither is used here).
//
// The reference's arg-max scan over a bin run ("first strictly greater |X|^2 wins", :957-979) is sequential and the
// runs are very uneven (1 .. ~140 bins at the top of a log view).  Here every run is cut into pieces of <= 16 bins
// (plan.cpp, MaxItem); a thread scans one piece with 16 independent LDS reads and merges through ds_max_u64 on the
// key (bits(|X|^2) << 32 | ~offset): larger square wins, equal squares -> smaller scan offset wins, which is
// exactly "first strictly greater".  Key 0 (no square > 0) falls back to `bin` like the reference's initial value.
template <int LR>
__device__ __forceinline__ float finishPixel(float val) { return finishMagnitude(val); }   // (late_fix.hpp)

template <int LR, int NT>
__device__ __forceinline__ void mapPixelsSerial(const StftParams &prm, const float *lds, int tid, long task)
{
#pragma clang fp contract(off)
    constexpr int R = 1 << LR, T = NT, N = R * R * R;
    const int total = int(prm.sides * prm.P);
    float *out = prm.mapped + size_t(task) * total;
    for (int idx = tid; idx < total; idx += T) {
        const PixelRec rec = prm.recs[idx];
        const int side = idx >= int(prm.P) ? 1 : 0;
        float val;
        if ((rec.kind & 1) == 0) {
            float acc = 0.f;
            int k = rec.a;
            for (int i = 0; i < rec.b; ++i) {
                const float m = lds[k + (k >> LR)];
                const float prod = m * prm.weights[rec.c + i];
                acc = acc + prod;
                k = (k == N) ? 0 : k + 1;
            }
            val = prm.invSize * acc;
        } else {
            float best = 0.f;
            int arg = rec.c;
            for (int i = 0; i < rec.b; ++i) {
                const int off = rec.a + i;
                const int k = side ? (N - off) : off;
                const float m = lds[k + (k >> LR)];
                const float sq = m * m + 0.f;                         // Math::square(csf[offset]) with imag == 0
                if (sq > best) { best = sq; arg = k; }
            }
            val = prm.invSize * lds[arg + (arg >> LR)];
        }
        out[idx] = finishPixel<LR>(val);
    }
}

// balanced version; `win` = one (k, bits(|X|^2)) pair per MaxItem in LDS.  One workgroup barrier and NO atomics
// (ds_max_u64 runs at only ~1 lane-op per 4-6 cycles: 3400 of them cost ~13k cycles per frame).
//   (a) every <=16-bin piece finds its local winner and stores the winner's bin and square;
//   (c) the pixel's thread replays the reference's scan ("first strictly greater", TransformDSP.inl:957-979) over its
//       pieces' winners, in scan order -> same arg-max, same ties.  All of a run's entries are fetched with
//       independent LDS reads first, so the replay is register arithmetic and its latency does not grow with the run.
// Table reads (items, records, tap weights) are issued in batches of independent loads: a thread's work list is
// tiny, so what matters is the number of dependent global-load round trips, not the byte count.
// The first batch of table reads is split off (prefetchTables / prefetchWeights) so that the kernel can issue it
// while the FFT's last barriers are still pending: the map phase then starts with its operands in registers.
// Where csf[k] lives in LDS.  WholeFrame: the fused kernel's bank-padded natural order, N = R^3.
template <int LR>
struct WholeFrameIndex {
    static constexpr int N = (1 << LR) * (1 << LR) * (1 << LR);
    static constexpr bool kSkipEmpty = false;        // both sides of a view fill the thread batches (measured: a skip gains nothing)
    // the fused kernel's csf array can be read as kMaxTaps + 1 CONTIGUOUS floats from a tap window's first entry: its pad slots
    // hold 0, csf[0 .. 9] is repeated behind csf[N] (the window's periodic indexing over N + 1 entries), and the plan's
    // weights11 table carries a zero where a window steps over a pad slot -- no per-tap address arithmetic, no per-tap select
    static constexpr bool kLinearTaps = true;
    __device__ __forceinline__ int size() const { return N; }
    __device__ __forceinline__ int operator()(int k) const { return k + (k >> LR); }
    __device__ __forceinline__ bool holds(int) const { return true; }
};
// OneSide (mapSideKernel, N = 2 R^3): LDS holds the csf range one side of the view touches, rotated by `off` (a multiple of
// 16, so the 16-aligned arg-max windows stay aligned and contiguous): k' = (k + off) mod (N + 1), same bank padding.
struct OneSideIndex {
    int n, off;
    static constexpr bool kSkipEmpty = true;         // one side's work list leaves whole waves of a batch slot empty: skip those
    static constexpr bool kLinearTaps = false;
    __device__ __forceinline__ int size() const { return n; }
    __device__ __forceinline__ int operator()(int k) const
    {
        int kp = k + off;
        kp = kp > n ? kp - (n + 1) : kp;
        return kp + (kp >> 5);
    }
    // is csf[k] inside the staged range (n/2 + 48 entries from csf index -off on)?
    __device__ __forceinline__ bool holds(int k) const
    {
        int kp = k + off;
        kp = kp > n ? kp - (n + 1) : kp;
        return kp >= 0 && kp < n / 2 + 48;
    }
};
// The slice of the plan's tables a workgroup maps: all of it (fused kernel) or one side's records and pieces.
struct MapView {
    const MaxItem *items;
    uint32_t nItems;          // pieces in this view
    uint32_t nItemsLeft;      // pieces [0, nItemsLeft) scan ascending k, the rest descending
    uint32_t itemBase;        // index of the view's first piece in the plan's list (PixelRec::kind counts from there)
    const PixelRec *recs;
    int total;                // records in this view
    int rightFrom;            // records [rightFrom, total) are right-side records (k = N - offset)
    float *out;               // [total]
    float *bestOut = nullptr; // optional (LDS): the winning SQUARE of the arg-max records [bestFrom, total) -> bestOut[idx - bestFrom], their pixel values -> bestOut[64 + ...], INSTEAD of out[]
    int bestFrom = 0;         // (spectrum_real.hip: the pixels whose run ends on csf[N/2] are settled after both channels are done)
};
__device__ __forceinline__ MapView wholeView(const StftParams &prm, long task)
{
    const int total = int(prm.sides * prm.P);
    return MapView{prm.items, prm.nItems, prm.nItemsLeft, 0u, prm.recs, total, int(prm.P), prm.mapped + size_t(task) * total};
}

template <int LR, int NT, typename Index = WholeFrameIndex<LR>>
struct MapPixelsBalanced {
    static constexpr int IB = 4;                                         // items per thread per batch (register budget)
    static constexpr int RB = 2;                                         // records per thread per batch (register budget)
    static constexpr int PB = 10;                                        // piece entries fetched per batch in (c)
    static constexpr int NW = Index::kLinearTaps ? kMaxTaps + 1 : kMaxTaps;   // weights (and taps) fetched per record
    uint32_t iw0[IB];
    PixelRec rec0[RB];
    float w0[RB][NW];

    __device__ __forceinline__ void loadItems(const MapView &v, uint32_t base, int tid, uint32_t (&iw)[IB]) const
    {
#pragma unroll
        for (int b = 0; b < IB; ++b) {
            const uint32_t it = base + b * NT + tid;
            iw[b] = it < v.nItems ? v.items[it].win : 0u;
        }
    }
    __device__ __forceinline__ void loadRecs(const MapView &v, int base, int tid, PixelRec (&rec)[RB]) const
    {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int idx = base + b * NT + tid;
            rec[b] = idx < v.total ? v.recs[idx] : PixelRec{2, 0, 0, 0};
        }
    }
    // tap weights: unconditional, independent loads (the weight table is padded by kMaxTaps zeros)
    __device__ __forceinline__ void loadWeights(const StftParams &prm, const PixelRec (&rec)[RB], float (&w)[RB][NW], int base, int tid,
                                                int total) const
    {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            if (Index::kLinearTaps) {                                    // fixed 11 floats per record, indexed by the record
                const int idx = base + b * NT + tid;
                // (arg-max records carry no weights: they all read record 0's slot -- one broadcast line instead of 44 bytes each)
                const float *src = prm.weights11 + size_t(idx < total && rec[b].kind == 0 ? idx : 0) * NW;
#pragma unroll
                for (int i = 0; i < NW; ++i) w[b][i] = src[i];
            } else {
                const int wbase = rec[b].kind == 0 ? rec[b].c : 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) w[b][i] = prm.weights[wbase + i];
            }
        }
    }
    __device__ __forceinline__ void prefetchTables(const MapView &v, int tid)
    {
        loadItems(v, 0u, tid, iw0);
        loadRecs(v, 0, tid, rec0);
    }
    __device__ __forceinline__ void prefetchWeights(const StftParams &prm, int tid, int total) { loadWeights(prm, rec0, w0, 0, tid, total); }

__device__ __forceinline__ void run(const StftParams &prm, const MapView &v, const Index at, const float *lds, float *win, int tid,
                                    long task)
{
#pragma clang fp contract(off)
    const int total = v.total;
    const int N = at.size();
    SGZ_CLK_HALF(true);
    float *out = v.out;
    // (a) arg-max pieces.  A piece is a 16-aligned window of csf (one 32-block of the padded layout, so its 16 floats are contiguous: one
    // base address + immediate offsets) with positions lo..hi valid.  The reference keeps the first offset whose SQUARE is strictly
    // greater (:957-979) and then reads csf there.  fl(m^2) is strictly increasing in |m| as long as the square is a normal float
    // (adjacent floats are >= 1.41 ulp(m^2) apart in m^2), so equal squares mean equal |m|: whichever of several tied offsets the scan
    // order picks, the VALUE is max |m| -- a plain float maximum (v_max3_f32 with |.| source modifiers), no squares, no positions, no
    // scan direction.  Runs whose maximum is too small for that argument (|m| < 2^-62: silence) are redone by the literal scan in (c).
    for (uint32_t base = 0; base < v.nItems; base += NT * IB) {
        uint32_t iw[IB];
        if (base == 0) {
#pragma unroll
            for (int b = 0; b < IB; ++b) iw[b] = iw0[b];
        } else loadItems(v, base, tid, iw);
        float mv[IB][16];
#pragma unroll
        for (int b = 0; b < IB; ++b) {
            if (Index::kSkipEmpty && base + b * NT + (uint32_t(tid) & ~63u) >= v.nItems) continue;   // wave-uniform
            const int k0 = int(iw[b] & 0xFFFFu) << 4;
            const float *src = lds + at(k0);
#pragma unroll
            for (int j = 0; j < 16; ++j) mv[b][j] = src[j];
        }
#pragma unroll
        for (int b = 0; b < IB; ++b) {
            if (Index::kSkipEmpty && base + b * NT + (uint32_t(tid) & ~63u) >= v.nItems) continue;
            const uint32_t it = base + b * NT + tid;
            const int lo = int((iw[b] >> 16) & 15u), hi = int((iw[b] >> 20) & 15u);
            const uint32_t mask = (0xFFFFu >> (15 - hi)) & (0xFFFFu << lo);   // valid positions: bits lo..hi
            float m[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)                                     // positions outside the run become +0, which can never win
                m[j] = __uint_as_float(__float_as_uint(mv[b][j]) & uint32_t(__builtin_amdgcn_sbfe(int(mask), j, 1)));
            float m5[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) m5[j] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(m[3 * j]), __builtin_fabsf(m[3 * j + 1])), __builtin_fabsf(m[3 * j + 2]));
            const float ma = __builtin_fmaxf(__builtin_fmaxf(m5[0], m5[1]), m5[2]);
            const float mb = __builtin_fmaxf(__builtin_fmaxf(m5[3], m5[4]), __builtin_fabsf(m[15]));
            if (it < v.nItems) win[it] = __builtin_fmaxf(ma, mb);
        }
    }
    SGZ_CLK(10);
    // (b) interpolated pixels (<= 10 taps, accumulated in tap order)
    const bool oneBatch = total <= NT * RB;                              // then the records stay in registers across the barrier
    PixelRec rec[RB];
    for (int base = 0; base < total; base += NT * RB) {
        float w[RB][NW], mv[RB][NW];
        if (base == 0) {
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                rec[b] = rec0[b];
#pragma unroll
                for (int i = 0; i < NW; ++i) w[b][i] = w0[b][i];
            }
        } else {
            loadRecs(v, base, tid, rec);
            loadWeights(prm, rec, w, base, tid, total);
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            if (Index::kSkipEmpty && base + b * NT + (tid & ~63) >= total) continue;
            int k = rec[b].kind == 0 ? rec[b].a : 0;
            if (Index::kLinearTaps) {
                const float *src = lds + at(k);                         // one address, immediate offsets
#pragma unroll
                for (int i = 0; i < NW; ++i) mv[b][i] = src[i];
            } else {
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    mv[b][i] = lds[at(k)];
                    k = (k == N) ? 0 : k + 1;
                }
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            if (Index::kSkipEmpty && base + b * NT + (tid & ~63) >= total) continue;
            const int idx = base + b * NT + tid;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const float prod = mv[b][i] * w[b][i];
                // taps accumulate in order (lanczosFilter restatement).  Linear form: entries that are not taps carry weight +0 and read
                // a finite value (a zeroed pad slot, or a bin of a frame that is finite wherever this pixel is), so they add +-0 to a
                // sum that is never -0: the same sum, bit for bit
                if (Index::kLinearTaps) acc = acc + prod;
                else acc = (i < rec[b].b) ? acc + prod : acc;
            }
            if (rec[b].kind == 0) out[idx] = finishPixel<LR>(prm.invSize * acc);
        }
    }
    SGZ_CLK(11);
    ldsBarrier();
    SGZ_CLK(12);
    // (c) resolve the arg-max pixels from their pieces' winners
    for (int base = 0; base < total; base += NT * RB) {
        if (!oneBatch) loadRecs(v, base, tid, rec);
        int first[RB], pieces[RB], maxPieces = 0;
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int idx = base + b * NT + tid;
            const bool right = idx >= v.rightFrom;
            first[b] = (rec[b].kind >> 1) - int(v.itemBase);
            // number of 16-aligned csf windows the run [a, a+b) spans (k = offset, or N - offset on the right side)
            const int kLo = right ? N - (rec[b].a + rec[b].b - 1) : rec[b].a;
            const int kHi = right ? N - rec[b].a : rec[b].a + rec[b].b - 1;
            pieces[b] = (rec[b].kind & 1) ? (kHi >> 4) - (kLo >> 4) + 1 : 0;
            maxPieces = pieces[b] > maxPieces ? pieces[b] : maxPieces;
        }
        float best[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) best[b] = 0.f;
        for (int p0 = 0; p0 < maxPieces; p0 += PB) {
            float e[RB][PB];
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int i = 0; i < PB; ++i) {
                    const int pc = p0 + i;
                    e[b][i] = win[pc < pieces[b] ? first[b] + pc : 0];
                    if (pc >= pieces[b]) e[b][i] = 0.f;
                }
#pragma unroll
            for (int b = 0; b < RB; ++b)
#pragma unroll
                for (int i = 0; i < PB; ++i) best[b] = __builtin_fmaxf(best[b], e[b][i]);
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int idx = base + b * NT + tid;
            if (!(rec[b].kind & 1)) continue;
            float val = best[b], bestSq = best[b] * best[b] + 0.f;          // Math::square(csf[offset]) of the winner (imag == 0)
            if (!(best[b] >= 0x1p-62f)) {
                // the run's squares are denormal, zero or NaN: distinct values can tie there -- the reference's scan, literally
                // (first strictly greater square in offset order; initial arg = bin, :953)
                const bool right = idx >= v.rightFrom;
                int arg = rec[b].c;
                bestSq = 0.f;
                for (int o = rec[b].a; o < rec[b].a + rec[b].b; ++o) {
                    const int k = right ? N - o : o;
                    const float mm = lds[at(k)];
                    const float sq = mm * mm + 0.f;
                    if (sq > bestSq) { bestSq = sq; arg = k; }
                }
                // (a right-side run without a positive square keeps arg = bin, a LEFT-side index (maxRBin = maxLBin = bin, :953): the
                // kernels that hold one side of csf per workgroup cannot read it and show 0 -- what the run's own bins say; with a real
                // FFT's rounding noise in a silent channel the reference does not get here either)
                val = at.holds(arg) ? lds[at(arg)] : 0.f;
            }
            const float pix = finishPixel<LR>(prm.invSize * val);
            if (v.bestOut && idx >= v.bestFrom) {
                // spectrum_real.hip: these pixels also depend on the other channel and are written once that is known -- (winning square,
                // pixel value) go to the workgroup's LDS, nothing is written here
                v.bestOut[idx - v.bestFrom] = bestSq;
                v.bestOut[64 + (idx - v.bestFrom)] = pix;
            } else out[idx] = pix;
        }
    }
}
};

// One workgroup = one (frame, pair).  LR = log2(R), N = R^3, T = R^2 threads of R points.
//
// Roles.  Pass 1: thread tid owns column t = tid.  Passes 2/3: a "slot" is 2R consecutive lanes (one wave at R = 32),
// slot s = tid / 2R, half h = (tid / R) & 1, l = tid % R:
//     h = 0:  q = s,                    index (t2, then q2) = l
//     h = 1:  q = R - s (R/2 if s = 0), index               = R-1-l
// so bin k = q + R q2 + T m3 of lane L and its mirror N-k = (R-q) + R (R-1-q2) + T (R-1-m3) sit in lanes L and L ^ R
// of the same wave, registers m3 and R-1-m3: the two-for-one split (TransformDSP.inl:858) needs one ds_bpermute per
// value and no LDS round trip.  Slot 0 holds q = 0 and q = R/2, which mirror onto themselves (other lane pattern);
// column 0 (q = 0, q2 = 0) mirrors inside thread 0 and is redone from a small LDS scratch by lanes 1..R/2-1.
// MIX = 0: Separate (re = L w, im = R w); MIX = 1: Left, Right, Merge, Side, MidSide; MIX = 2: Complex (the window of MIX = 0).
// MIX != 0 also redoes the pixels that reach a csf entry the reference leaves complex (complex_dc.hpp).
// FULLW: W == N (no zero padding): plain global loads.  Measured on gfx950 (tools/ubench/stream.hip): a workgroup's
// 3R strided dword loads complete in ~7-8.6 k clocks as global loads and in ~13 k as raw buffer loads, so the buffer
// form (whose out-of-range reads return 0 = the zero padding of prepareTransform, :220-223) is kept for W < N only.
// bid / nb: index of this workgroup among the nb workgroups of the launch; lds: the dynamic LDS of the launch.
//
// HALF = 0 / 1: the workgroup transforms one half of a 2N-point frame (N = R^3 here), split by decimation in frequency:
//     X[2j]   = FFT_N( z[n] + z[n + N] )[j]                 (HALF = 0)
//     X[2j+1] = FFT_N( (z[n] - z[n + N]) W_2N^n )[j]        (HALF = 1)
// The even half mirrors like a whole frame (X[2N - 2j] is its own bin N - j).  In the odd half bin j pairs with N-1-j
// (2N - (2j+1) = 2(N-1-j) + 1): all three digits are complemented, so with q = R-1-s in the upper half-waves the partner is
// lane L ^ R everywhere and there is no self-mirroring column.  W_2N^n for n = t + T a factors into the constant
// W_2R^a (a = register, before pass 1) and W_2N^t, which rides on the pass-1 twiddles (TwFactors<LR, true>).  The
// magnitudes go to HBM (csf of the 2N-point frame, element 2j + HALF); spectrum_generic.hip's genericMap maps them.
// ZOUT: stop after pass 3 and write the raw transform Z to prm.zOut as two natural-order arrays, re[N] then im[N] (Phase mode:
// its bins stay complex and its split / map are HBM-resident kernels).
template <int LR, int MIX, bool FULLW, int HALF = -1, bool ZOUT = false, bool WCOS = false>
__device__ __forceinline__ void stftMapBody(const StftParams &prm, float *lds, const long bid, const long nb)
{
    constexpr int R = 1 << LR;
    constexpr int T = R * R;
    constexpr int N = R * T;
    constexpr int PADSTRIDE = T + (T >> LR);          // padded distance between k and k + T
    constexpr int WRAP = N + (N >> LR) + 1;           // csf[0 .. 9] again, behind csf[N] (WholeFrameIndex::kLinearTaps)
    constexpr int SCRATCH = N + (N >> LR) + 12;       // float index of column 0's 2R-float scratch
    constexpr int SPEC = SCRATCH + 2 * R + 4;         // float index of the kSpecBins csf entries that stay complex (complex_dc.hpp)
    constexpr int SLOTS = SPEC + 2 * kSpecBins;       // float index (even) of the arg-max piece winners (nItems uint2)
    constexpr int TILE = R * (R + 1);
    const int tid = threadIdx.x;
    SGZ_CLK_HALF(HALF < 0 || HALF == int((prm.ablate >> 15) & 1u));   // debug clocks: which half reports
    const long tasks = prm.frames * long(prm.C) * (HALF >= 0 ? 2 : 1);
    const int slot = tid >> (LR + 1), half = (tid >> LR) & 1, l = tid & (R - 1);
    const int q = half ? (HALF == 1 ? R - 1 - slot : (slot == 0 ? R / 2 : R - slot)) : slot;
    const int ix = half ? R - 1 - l : l;                                // t2 in pass 2, q2 in pass 3
    const bool split = (prm.sides == 2);
    const int mode = prm.mode;
    float *win = lds + SLOTS;                                           // one |csf| maximum per arg-max piece
    const bool balanced = prm.items != nullptr;

    // XCD-aware task order.  Workgroup b is observed to run on XCD b % 8 (a speed assumption only, never a
    // correctness one).  The workgroups run in rounds of `roundSize` (= number of CUs: one workgroup fills a CU); inside a
    // round XCD x gets a contiguous range of frames, so that the workgroups sharing an L2 walk consecutive
    // (75 %-overlapping) frames together and each sample is fetched from HBM once per XCD; and round r covers frames
    // [r * roundSize, ...), so that frames complete roughly in time order.  One workgroup per task, no persistent frame loop (LLVM would hoist and spill addresses).
    long task = bid;
    if (tasks >= 64 && prm.roundSize >= 8 && prm.roundSize % 8 == 0) {
        const long base = (bid / prm.roundSize) * prm.roundSize;
        const long nbr = nb - base < long(prm.roundSize) ? nb - base : long(prm.roundSize);
        const long x = (bid - base) % 8, i = (bid - base) / 8;
        const long per = nbr / 8, extra = nbr % 8;                     // XCD x owns per + (x < extra) workgroups of this round
        task = base + x * per + (x < extra ? x : extra) + i;
    }
    if (HALF >= 0) {
        if (int(task & 1) != HALF) return;                             // wave-uniform: the kernel wrapper calls both bodies
        task >>= 1;
    }
    if (prm.C > 1) {
        // several pairs: walk the launch pair-major, so that the workgroups that share an L2 transform consecutive
        // (overlapping) frames of the same pair rather than the same frame of unrelated pairs
        const long pr = task / prm.frames, fr = task - pr * prm.frames;
        task = fr * prm.C + pr;
    }
    MapPixelsBalanced<LR, T> mapper;
    const bool doMap = HALF < 0 && !ZOUT && balanced && prm.mapped && !SGZ_ABLATED(16u);
    SGZ_CLK(0);
    SGZ_WCLK(0);
    if (prm.binsIn == nullptr) {
        v2 c[R];                                                       // (re, im) pairs: packed fp32 butterflies (fft_common.hpp)
        {
            // ---------------------------------------------------------------- load + window + channel mix
            // strided dword loads; reads past W return 0 = the zero padding of prepareTransform (:220-223)
            const long gtask = task + prm.taskBase;
            const long frame = gtask / prm.C;
            const int pair = int(gtask - frame * prm.C);
            const float *L = prm.planar + size_t(2 * pair) * prm.chStride + size_t(frame) * prm.hop;
            // raw samples land in scalar registers; the (re, im) pairs are formed by the window multiply
            float lv[R], rv[R], w[R];
            if (WCOS) {                                            // (FULLW, whole frame: spectrum_fft.hip instantiates it that way)
                const float *Rp = L + prm.chStride;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const uint32_t off = uint32_t(tid + j * T) * 4u;
                    lv[j] = ldg(L, off);
                    rv[j] = ldg(Rp, off);
                }
                const float2 ph = ldg(prm.winPhase, uint32_t(tid) * 8u);
#pragma unroll
                for (int j = 0; j < R; ++j) {                          // cos(phi + 2 pi j / R) by the compile-time rotation
                    constexpr int S32 = 32 / R;
                    const int a = (j * S32) % 32;
                    const float cj = a <= 16 ? cos32(a) : cos32(32 - a), sj = a <= 16 ? sin32(a) : -sin32(32 - a);
                    w[j] = prm.winP0 + prm.winP1 * (ph.x * cj - ph.y * sj);
                }
            } else if (FULLW) {
                const float *Rp = L + prm.chStride;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const uint32_t off = uint32_t(tid + j * T) * 4u;
                    lv[j] = ldg(L, off);
                    rv[j] = ldg(Rp, off);
                    w[j] = ldg(prm.window, off);
                }
            } else {
                const __amdgpu_buffer_rsrc_t rsL = makeRsrc(L, prm.W * 4u);
                const __amdgpu_buffer_rsrc_t rsR = makeRsrc(L + prm.chStride, prm.W * 4u);
                const __amdgpu_buffer_rsrc_t rsW = makeRsrc(prm.window, prm.W * 4u);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    lv[j] = bufLoad(rsL, tid * 4, j * (T * 4));
                    rv[j] = bufLoad(rsR, tid * 4, j * (T * 4));
                    w[j] = bufLoad(rsW, tid * 4, j * (T * 4));
                }
            }
            // prepareTransform channel mixes (TransformDSP.inl:59-216): `l * w`, `r * w`, `(l +- r) * w * 0.5f`.  The channel a mode
            // does not use is SELECTED away, not multiplied by 0 (an Inf / NaN sample in it must not reach the frame, as in the
            // reference, which never reads it); x + 0 and x * 1 are exact, so the used channel rounds like the reference's expression.
            const bool oneCh = mode == SGZ_CH_LEFT || mode == SGZ_CH_RIGHT, midSide = mode == SGZ_CH_MIDSIDE;
            const float mixSgn = mode == SGZ_CH_SIDE ? -1.f : 1.f, mixS = oneCh ? 1.f : 0.5f;
            auto windowed = [&](float lx, float rx, float wx) {
                if (MIX != 1) return v2{lx * wx, rx * wx};
                const float a = mode == SGZ_CH_RIGHT ? rx : lx, b = oneCh ? 0.f : rx;
                const float il = midSide ? lx : 0.f, ir = midSide ? rx : 0.f;
                return v2{(a + mixSgn * b) * wx * mixS, (il - ir) * wx * mixS};
            };
#pragma unroll
            for (int j = 0; j < R; ++j) c[j] = windowed(lv[j], rv[j], w[j]);
            if (HALF >= 0) {
                // (the empty asm pins each product above the next batch of loads: instruction selection would otherwise
                // sink the multiplies to their first use, below all 6R loads, and spill the raw samples)
#pragma unroll
                for (int j = 0; j < R; ++j) asm volatile("" : "+v"(c[j]));
                // second half of the 2N-point window, in two batches of R/2 columns (128-VGPR budget): z[n] +- z[n + N]
#pragma unroll
                for (int batch = 0; batch < 2; ++batch) {
                    __builtin_amdgcn_sched_barrier(0);
                    float l2[R / 2], r2[R / 2], w2[R / 2];
                    if (FULLW) {
                        const float *Rp = L + prm.chStride;
#pragma unroll
                        for (int jj = 0; jj < R / 2; ++jj) {
                            const uint32_t off = uint32_t(tid + (batch * (R / 2) + jj) * T + N) * 4u;
                            l2[jj] = ldg(L, off);
                            r2[jj] = ldg(Rp, off);
                            w2[jj] = ldg(prm.window, off);
                        }
                    } else {
                        const __amdgpu_buffer_rsrc_t rsL = makeRsrc(L, prm.W * 4u);
                        const __amdgpu_buffer_rsrc_t rsR = makeRsrc(L + prm.chStride, prm.W * 4u);
                        const __amdgpu_buffer_rsrc_t rsW = makeRsrc(prm.window, prm.W * 4u);
#pragma unroll
                        for (int jj = 0; jj < R / 2; ++jj) {
                            const int so = ((batch * (R / 2) + jj) * T + N) * 4;
                            l2[jj] = bufLoad(rsL, tid * 4, so);
                            r2[jj] = bufLoad(rsR, tid * 4, so);
                            w2[jj] = bufLoad(rsW, tid * 4, so);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < R / 2; ++jj) {
                        const int j = batch * (R / 2) + jj;
                        const v2 z = windowed(l2[jj], r2[jj], w2[jj]);
                        c[j] = HALF ? c[j] - z : c[j] + z;
                        asm volatile("" : "+v"(c[j]));
                    }
                }
                if (HALF == 1) {
                    // times W_2R^a (the register part of W_2N^n, n = t + T a)
#pragma unroll
                    for (int a = 1; a < R; ++a) {
                        constexpr int S = 64 / (2 * R);                // index step into the W_64 table
                        if (a * S == 16) c[a] = v2{c[a].y, -c[a].x};   // times -i
                        else c[a] = cmulConjK(c[a], v2{cos64(a * S), sin64(a * S)});
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the twiddle loads below the 3R sample loads (128-VGPR budget)
        SGZ_CLK(13);
        // ---------------------------------------------------------------------- pass 1
        if (!SGZ_ABLATED(1u)) ditPacked<LR, 0>(c);
        SGZ_CLK(14);
        if (!SGZ_ABLATED(32u)) {
            TwFactors<LR, HALF == 1> tw;
            tw.load(HALF == 1 ? prm.tw1odd : prm.tw1, tid, T);
            tw.apply(c);                                               // times W_N^{t q} (odd half: W_2N^{t (2q+1)})
        }
        SGZ_CLK(1);
        // -------------------------------------------------------------- exchange 1 (workgroup-wide)
        // (re, im) pairs move as 64-bit LDS operations: measured 6.0 clocks per ds_write_b64 and 2.3 per ds_read_b64 against
        // 2 x 4.1 and 2 x 2.1 for the dword forms (tools/ubench/lds.hip).  N complex values are 2x the LDS, so the lower and
        // the upper half of the workgroup take turns as writers; everybody reads its R/2 values of the first round into
        // spare registers (the twiddles are dead, the map tables not loaded yet).
        if (!SGZ_ABLATED(2u)) {
            v2 *lds2 = reinterpret_cast<v2 *>(lds);
            const int rd = q * (T / 2) + ix;
            v2 lo[R / 2];
            if (tid < T / 2) {
#pragma unroll
                for (int qq = 0; qq < R; ++qq) lds2[qq * (T / 2) + tid] = c[brev(qq, LR)];
            }
            ldsBarrier();
#pragma unroll
            for (int j2 = 0; j2 < R / 2; ++j2) lo[j2] = lds2[rd + R * j2];
            ldsBarrier();
            if (tid >= T / 2) {
#pragma unroll
                for (int qq = 0; qq < R; ++qq) lds2[qq * (T / 2) + tid - T / 2] = c[brev(qq, LR)];
            }
            ldsBarrier();
#pragma unroll
            for (int j2 = 0; j2 < R / 2; ++j2) {
                c[j2 + R / 2] = lds2[rd + R * j2];
                c[j2] = lo[j2];
            }
        }
        // Exchange 2's tiles alias exchange 1's array, so every wave must have finished READING exchange 1 before any wave writes a
        // tile.  The barrier sits here, before pass 2, not after it: from here to the |X| barrier a wave depends on nobody, so a wave
        // that is done with pass 2 goes straight into its (LDS-bound) transposes while the other waves of its SIMD are still in their
        // (VALU-bound) butterflies -- with the barrier after pass 2 all sixteen waves entered the LDS phase together.
        ldsBarrier();
        SGZ_CLK(2);
        // ---------------------------------------------------------------------- pass 2 (t2 = ix)
        if (!SGZ_ABLATED(1u)) ditPacked<LR, 0>(c);
        if (!SGZ_ABLATED(32u)) {
            TwFactors<LR> tw;
            tw.load(prm.tw2, ix, R);
            tw.apply(c);                                               // times W_T^{t2 q2}
        }
        SGZ_CLK(3);
        // ----------------------------------- exchange 2: R x R transposes inside each R-lane group (wave-local tiles)
        if (!SGZ_ABLATED(4u)) {
            const int tile = q * TILE;
#pragma unroll
            for (int q2 = 0; q2 < R; ++q2) lds[tile + q2 * (R + 1) + ix] = c[brev(q2, LR)].x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int j = 0; j < R; ++j) c[j].x = lds[tile + ix * (R + 1) + j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q2 = 0; q2 < R; ++q2) lds[tile + q2 * (R + 1) + ix] = c[brev(q2, LR)].y;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int j = 0; j < R; ++j) c[j].y = lds[tile + ix * (R + 1) + j];
        }
        SGZ_CLK(4);
        // ---------------------------------------------------------------------- pass 3 (q2 = ix)
        if (!SGZ_ABLATED(1u)) ditPacked<LR, 0>(c);
        SGZ_CLK(5);
        SGZ_WCLK(1);
        if (ZOUT) {
            // plane by plane through the csf layout (k + k / R): two natural-order arrays, re[N] then im[N], coalesced
            const int kcz = q + R * ix, bz = kcz + (kcz >> LR);
            float *dst = reinterpret_cast<float *>(prm.zOut + size_t(task) * N);
#pragma unroll
            for (int plane = 0; plane < 2; ++plane) {
                ldsBarrier();                                       // exchange-2 tiles / the previous plane are dead
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) lds[bz + m3 * PADSTRIDE] = plane ? c[brev(m3, LR)].y : c[brev(m3, LR)].x;
                ldsBarrier();
#pragma unroll 8
                for (int k = tid; k < N; k += T) dst[plane * N + k] = lds[k + (k >> LR)];
            }
            return;
        }
        // Z[kc + T m3] at register brev(m3),  kc = q + R q2.  The magnitudes go to their own scalar registers so that the
        // (re, im) pairs die as the mirror consumes them (a half-dead 64-bit tuple still holds two registers).
        float mag[R];
        const int kc = q + R * ix;
        const int base = kc + (kc >> LR);                              // padded LDS address of k = kc
        if (split && !SGZ_ABLATED(8u)) {
            if (HALF != 1 && tid == 0) {                               // column 0 mirrors onto itself: redone below
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) {
                    lds[SCRATCH + 2 * m3] = c[brev(m3, LR)].x;
                    lds[SCRATCH + 2 * m3 + 1] = c[brev(m3, LR)].y;
                }
            }
            // lane holding Z[N - k]: L ^ R, except in slot 0 (q = 0: q2' = R - q2 ; q = R/2: q2' = R-1-q2, same half)
            const int lane = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
            int plane = lane ^ R;
            if (HALF != 1 && slot == 0) plane = (lane & ~(R - 1)) | (half ? R - 1 - l : ((R - l) & (R - 1)));
            plane <<= 2;
            auto partner = [&](float v) {
                return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plane, __builtin_bit_cast(int, v)));
            };
#pragma unroll
            for (int m3 = 0; m3 < R / 2; ++m3) {
                const int ia = brev(m3, LR), ib = brev(R - 1 - m3, LR);          // k < N/2 at ia, k > N/2 at ib
                const v2 ma = v2{partner(c[ib].x), partner(c[ib].y)};            // Z[N-k] for the bin at ia
                const v2 mb = v2{partner(c[ia].x), partner(c[ia].y)};            // Z[N-k] for the bin at ib
                // k < N/2: X1 = (Z[k] + conj Z[N-k])/2 ; k > N/2: X2 = (Z[N-k] - conj Z[k])/(2i)  (magnitudes only)
                const float ua = c[ia].x + ma.x, va = c[ia].y - ma.y, ub = c[ib].x - mb.x, vb = c[ib].y + mb.y;
                mag[ia] = 0.5f * __builtin_amdgcn_sqrtf(ua * ua + va * va);
                mag[ib] = 0.5f * __builtin_amdgcn_sqrtf(ub * ub + vb * vb);
            }
        } else {
            if (HALF != 1 && tid == 0) { lds[SCRATCH] = c[0].x; lds[SCRATCH + 1] = c[0].y;
                            lds[SCRATCH + R] = c[brev(R / 2, LR)].x; lds[SCRATCH + R + 1] = c[brev(R / 2, LR)].y; }
            if ((MIX != 0 || HALF >= 0) && (HALF < 0 || prm.dcOut)) {     // (halves: MIX = 0 also serves Complex)
                // The csf entries the reference leaves complex (complex_dc.hpp), with its scale factor: to LDS for this
                // kernel's own redo, or (halves) to the task's slots in HBM.  Frame bin N_f - 8 + s lives at kc = T - 8 + s,
                // m3 = R - 1; N_f / 2 + s at kc = s, m3 = R / 2 (a half owns every other one: bin 2 j + HALF <-> its j).
                auto put = [&](int s, v2 z, float f) {
                    if (HALF < 0) { lds[SPEC + 2 * s] = f * z.x; lds[SPEC + 2 * s + 1] = f * z.y; }
                    else prm.dcOut[size_t(task) * kSpecBins + s] = make_float2(f * z.x, f * z.y);
                };
                if (mode == SGZ_CH_COMPLEX) {
                    if (HALF <= 0 && tid == 0) put(0, c[0], 0.5f);
                } else {
                    constexpr int WR = HALF >= 0 ? 4 : 8;              // entries of this workgroup on either side
                    if (ix == R - 1 && q >= R - WR) put(HALF >= 0 ? 2 * (q - (R - WR)) + HALF : q - (R - WR), c[R - 1], 1.f);
                    if (ix == 0 && q < WR) {
                        const int s = HALF >= 0 ? 2 * q + HALF : q;
                        put(8 + s, c[brev(R / 2, LR)], s == 0 ? 0.5f : 1.f);
                    }
                }
            }
#pragma unroll
            for (int m3 = 0; m3 < R; ++m3) {                           // csf[k] = |Z[k]| (TransformDSP.inl:553-560, :993-1002)
                const int i = brev(m3, LR);
                mag[i] = __builtin_amdgcn_sqrtf(c[i].x * c[i].x + c[i].y * c[i].y);
            }
        }
        SGZ_CLK(6);
        if (doMap) mapper.prefetchTables(wholeView(prm, task), tid);   // im[] is dead: its registers take the map tables
        // csf[N/2-1] *= 0.5 (quirk Q3, :864); of a 2N-point frame that bin is the odd half's j = N/2 - 1
        if (HALF != 0 && split && q == R - 1 && ix == R - 1) mag[brev(R / 2 - 1, LR)] *= 0.5f;
        ldsBarrier();                                               // exchange-2 tiles are dead: M may overwrite them
                                                                       // (measured: moving this barrier up behind the tile reads, as was
                                                                       // done for exchange 1, costs 7 % on a tail-free launch)
#pragma unroll
        for (int m3 = 0; m3 < R; ++m3) lds[base + m3 * PADSTRIDE] = mag[brev(m3, LR)];
        if (HALF < 0) {                                             // WholeFrameIndex::kLinearTaps: see the index policy
            lds[(tid + 1) * (R + 1) - 1] = 0.f;                     // the T = N / R pad slots
            if (kc >= 1 && kc <= 9) lds[WRAP + kc] = mag[brev(0, LR)];
        }
        // Column 0 (k = T m3, all held by thread 0) mirrors onto itself and DC / Nyquist are special: redone from thread 0's
        // scratch copy by lanes of the SAME wave, after that wave's own stores above (one wave's LDS operations execute in
        // order), so no extra workgroup barrier is needed.
        if (HALF != 1 && tid < R / 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (split && tid >= 1) {
                const int m3 = tid;                                    // k = T m3 pairs with T (R - m3)
                const float ar = lds[SCRATCH + 2 * m3], ai = lds[SCRATCH + 2 * m3 + 1];
                const float br = lds[SCRATCH + 2 * (R - m3)], bi = lds[SCRATCH + 2 * (R - m3) + 1];
                const float ua = ar + br, va = ai - bi, ub = br - ar, vb = bi + ai;
                lds[m3 * PADSTRIDE] = 0.5f * __builtin_amdgcn_sqrtf(ua * ua + va * va);
                lds[(R - m3) * PADSTRIDE] = 0.5f * __builtin_amdgcn_sqrtf(ub * ub + vb * vb);
            }
            if (tid == 0) {
                const float dcRe = lds[SCRATCH], dcIm = lds[SCRATCH + 1];
                const float nyRe = lds[SCRATCH + R], nyIm = lds[SCRATCH + R + 1];      // m3 = R/2
                if (split) {
                    lds[N + (N >> LR)] = dcIm * 0.5f;                // csf[N]   = Im(csf[0]) * 0.5   (TransformDSP.inl:861)
                    lds[0] = dcRe * 0.5f;                            // csf[0]   = Re(csf[0]) * 0.5   (:862)
                    lds[WRAP] = dcRe * 0.5f;
                    lds[N / 2 + ((N / 2) >> LR)] = 0.5f * __builtin_amdgcn_sqrtf(nyRe * nyRe + nyIm * nyIm);   // :863
                } else {
                    lds[N + (N >> LR)] = 0.f;
                    lds[0] = 0.5f * __builtin_amdgcn_sqrtf(dcRe * dcRe + dcIm * dcIm);
                    lds[WRAP] = lds[0];
                    if (mode != SGZ_CH_COMPLEX)
                        lds[N / 2 + ((N / 2) >> LR)] = 0.5f * __builtin_amdgcn_sqrtf(nyRe * nyRe + nyIm * nyIm);
                }
            }
        }
        if (doMap) mapper.prefetchWeights(prm, tid, int(prm.sides * prm.P));
        ldsBarrier();
    } else {
        // test path (sgz_stage_map_from_bins): csf magnitudes come from HBM
        const float *src = prm.binsIn + size_t(task) * (N + 1);
        if (doMap) { mapper.prefetchTables(wholeView(prm, task), tid); mapper.prefetchWeights(prm, tid, int(prm.sides * prm.P)); }
        for (int k = tid; k <= N; k += T) lds[k + (k >> LR)] = src[k];
        lds[(tid + 1) * (R + 1) - 1] = 0.f;
        if (tid < 10) lds[WRAP + tid] = src[tid];
        ldsBarrier();
    }
    SGZ_CLK(7);

    if (HALF >= 0) {
        // csf of the 2N-point frame: this half owns the elements 2j + HALF (and the even half csf[2N])
        // (binsSplit: as two contiguous arrays for mapSideKernel -- even bins, csf[2N], odd bins -- instead of interleaved)
        float *dst = prm.binsOut + size_t(task) * (2 * N + 1) + (prm.binsSplit ? HALF * (N + 1) : HALF);
        const int stride = prm.binsSplit ? 1 : 2;
#pragma unroll 8
        for (int k = tid; k < N; k += T) dst[stride * k] = lds[k + (k >> LR)];
        if (HALF == 0 && tid == 0) dst[stride * N] = lds[N + (N >> LR)];
        SGZ_CLK(8);
        return;
    }
    if (prm.binsOut) {
        float *dst = prm.binsOut + size_t(task) * (N + 1);
        for (int k = tid; k <= N; k += T) dst[k] = lds[k + (k >> LR)];
    }
    SGZ_CLK(8);
    SGZ_WCLK(2);
    // -------------------------------------------------------------------------- pixel mapping
    if (prm.mapped && !SGZ_ABLATED(16u)) {
        if (balanced) mapper.run(prm, wholeView(prm, task), WholeFrameIndex<LR>{}, lds, win, tid, task);
        else mapPixelsSerial<LR, T>(prm, lds, tid, task);
    }
    if (MIX != 0 && prm.nDcPixels != 0 && prm.mapped && prm.binsIn == nullptr && !SGZ_ABLATED(16u)) {
        __syncthreads();                                               // the pixels' first values are written by other threads
        float *out = prm.mapped + size_t(task) * (prm.sides * prm.P);
        for (uint32_t i = tid; i < prm.nDcPixels; i += T) {
            const uint32_t x = prm.dcPixels[i];
            out[x] = complexDcPixel(prm.recs[x], prm.weights, prm.invSize, N, uint32_t(mode), [&](int k) { return lds[k + (k >> LR)]; },
                                    [&](int s) { return make_float2(lds[SPEC + 2 * s], lds[SPEC + 2 * s + 1]); });
        }
    }
    SGZ_CLK(9);
    SGZ_WCLK(3);
}

}  // namespace sgz
