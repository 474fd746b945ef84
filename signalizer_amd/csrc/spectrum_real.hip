// spectrum_real.hip -- K_A, channel-split form: one workgroup per (frame, pair, CHANNEL).  gfx950 only.
//
// SpectrumChannels::Separate transforms z = L w + i R w with one N-point complex FFT and splits the result (TransformDSP.inl:858-869).
// The two channels never meet again except in one bin (csf[N/2], see below): left pixels read csf[0 .. N/2], right pixels
// csf[N/2 .. N].  So the frame is cut along that line: each channel gets its own workgroup, which computes the channel's spectrum
// with a REAL-input FFT -- M = N/2 complex points z[n] = x[2n] + i x[2n+1], then X[k] = E[k] + W_N^k O[k] -- keeps the M + 1
// magnitudes of its side of csf in LDS and maps its side's pixels.  Same flops as the two-for-one transform, but:
//   * N = 32768: a task is 512 threads x <= 128 VGPRs and <= 80 KB of LDS, so TWO tasks share a CU.  They are in different phases
//     (one in its LDS-bound transposes while the other is in its VALU-bound butterflies), which is the overlap the sixteen
//     lock-stepped waves of the whole-frame kernel could not find among themselves; and the work list is twice as fine (696 tasks
//     on 512 slots instead of 348 on 256), so the partial last round of workgroups costs half as much;
//   * N = 65536: the transform is the R^3 = 32768-point in-register FFT itself, fused with the map -- no half-frame workgroups, no
//     csf round trip through HBM, no second kernel.
// Layout: M = R1 R^2 complex points, R = 32, R1 = 16 (N = 32768) or 32 (N = 65536); T = R1 R threads of R points.
//   pass 1  thread t owns the columns c = t + T u (u < R / R1), R1 points z[c + R^2 j] each: radix-R1 DIF, times W_M^{c q1}
//   exch 1  workgroup-wide, 64-bit LDS operations, two rounds of R1 x 512 values
//   pass 2  role (q1, c_lo): radix-R DIF over c_hi, times W_{R^2}^{c_lo q2}
//   exch 2  R x R transposes inside each R-lane group (wave-local)
//   pass 3  role (q1, q2): radix-R DIF over c_lo -> Z[q1 + R1 q2 + T m3]
//   recombination: Z[M - k] sits in lane L ^ R, register R-1-m3 (same construction as stft_body.hpp) ->
//           X[k] = ( Z[k] + conj Z[M-k] ) / 2  -  i W_N^k ( Z[k] - conj Z[M-k] ) / 2 ,   |X[k]| -> LDS (this side's csf order)
//   map     MapPixelsBalanced (stft_body.hpp) on this side's records and arg-max pieces.
// csf[N/2] is the one entry that mixes the channels: the reference halves the PACKED bin there, |X_L[M] + i X_R[M]| / 2
// (TransformDSP.inl:863).  It is the last offset of either side's arg-max scan and is compared with a strict >, so nobody waits for
// it: a workgroup maps with 0 in its place, publishes its own Nyquist bin and the winning squares of the (top) pixels whose run ends
// there, then raises its epoch flag; the channel that finishes second sees the other's flag and settles those pixels for both
// sides (store own flag, load the partner's, both sequentially consistent: at least one of the two sees the other).
// Everything else the path needs (mono modes, Complex, Phase, zero-padded windows, views whose filter taps wrap around csf) stays
// on the whole-frame kernels; plan.cpp decides (Plan::realSplit).
#include <algorithm>

#include "stft_body.hpp"

#ifdef SGZ_DEBUG
#define RCLK(slot)                                                                                                     \
    do {                                                                                                               \
        if (prm.phaseClock && (tid & 63) == 0 && unit == long(prm.clkUnit)) prm.phaseClock[16 * (tid >> 6) + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define RCLK(slot) do { } while (0)
#endif

namespace sgz {

// csf index -> LDS float index of this side's array: left holds csf[0 .. M], right csf[M .. N], both at positions 0 .. M
struct ChannelIndex {
    int n, off;
    static constexpr bool kSkipEmpty = true;
    static constexpr bool kLinearTaps = false;
    __device__ __forceinline__ int size() const { return n; }
    __device__ __forceinline__ int operator()(int k) const { const int i = k - off; return i + (i >> 5); }
    __device__ __forceinline__ bool holds(int k) const { return k >= off && k <= off + n / 2; }
};

// |X[k]| of the real-input transform from a = Z[k], b = Z[M - k] and w = W_N^k = (cos, -sin):
//   2 X = (a + conj b) - i w (a - conj b)
__device__ __forceinline__ float realBinMag(v2 a, v2 b, v2 w)
{
    const float ex = a.x + b.x, ey = a.y - b.y;          // a + conj b
    const float dx = a.x - b.x, dy = a.y + b.y;          // a - conj b
    // -i w d = -i (w.x + i w.y)(dx + i dy) = (w.x dy + w.y dx) + i (w.y dy - w.x dx)
    const float xr = ex + (w.x * dy + w.y * dx);
    const float xi = ey + (w.y * dy - w.x * dx);
    return 0.5f * __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
}

// the R1-point DIFs of a thread's U = R / R1 columns (registers [u R1, (u + 1) R1))
template <int R, int R1, int U, int u = 0>
__device__ __forceinline__ void pass1Columns(v2 (&c)[R])
{
    if constexpr (u < U) {
        difPacked<R, R1, u * R1>(c);
        pass1Columns<R, R1, U, u + 1>(c);
    }
}

// MONO: SpectrumChannels Left / Right / Merge / Side -- ONE real signal per (frame, pair) (the reference transforms it as a complex frame
// with a zero imaginary part, TransformDSP.inl:59-135): one workgroup per task, no pair exchange.  csf[0] = |X[0]| / 2 and
// csf[N/2] = X[N/2] / 2 (:547-552; the latter stays signed: the reference leaves it complex, and X[N/2] of a real signal is real).
template <int LR1, bool WCOS, int MIX = 0>                 // MIX: 0 Separate (two channel workgroups), 1 mono Left / Right, 2 MidSide (two workgroups on mid and side), 3 mono Merge / Side
__global__ void __launch_bounds__(1 << (LR1 + 5), 4) stftRealKernel(const RealParams prm)
{
    constexpr int LR = 5, R = 32, R1 = 1 << LR1, T = R1 * R, RR = R * R, M = R1 * RR, N = 2 * M, U = R / R1;
    constexpr bool MONO = MIX == 1 || MIX == 3;
    constexpr bool mixed = MIX >= 2;                    // the signal is (l +- r) / 2: compile-time, the second channel's loads cost registers
    constexpr int PADSTRIDE = T + (T >> 5);             // padded distance between k and k + T
    constexpr int TILE = R * (R + 1);
    constexpr int XFLOATS = ((M + 1) + ((M + 1) >> 5) + 2) & ~1;      // this side's |X| array (padded) -- the winners follow it
    constexpr int SCRATCH = XFLOATS;                    // column 0's 2R floats live where the winners will be written later
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int slot = tid >> 6, half = (tid >> 5) & 1, l = tid & 31, group = tid >> 5;
    const int q1 = half ? (slot == 0 ? R1 / 2 : R1 - slot) : slot;
    const int ix = half ? R - 1 - l : l;                // c_lo in pass 2, q2 in pass 3

    // ---- work list: unit = (frame, pair, channel); XCD-aware order as in stft_body.hpp (a speed assumption only)
    const long units = prm.frames * long(prm.C) * (MONO ? 1 : 2);
    long unit = blockIdx.x;
    if (units >= 64 && prm.roundSize >= 8 && prm.roundSize % 8 == 0) {
        const long nb = gridDim.x, bid = blockIdx.x;
        const long base = (bid / prm.roundSize) * prm.roundSize;
        const long nbr = nb - base < long(prm.roundSize) ? nb - base : long(prm.roundSize);
        const long x = (bid - base) % 8, i = (bid - base) / 8;
        const long per = nbr / 8, extra = nbr % 8;
        unit = base + x * per + (x < extra ? x : extra) + i;
    }
    const int side = MONO ? 0 : int(unit & 1);
    long task = MONO ? unit : unit >> 1;                // (frame, pair)
    if (prm.C > 1) { const long pr = task / prm.frames, fr = task - pr * prm.frames; task = fr * prm.C + pr; }
    const long frame = task / prm.C;
    const int pair = int(task - frame * prm.C);
    const long partner = (task << 1) | (side ^ 1);      // ny / flag / best slots are indexed by task * 2 + side
    const long self = (task << 1) | side;

    __shared__ float sLate[128];                                            // own late pixels: winning squares [0, 64), pixel values [64, 128)
    __shared__ float sNyOwn;
    // map tables of this side
    const uint32_t nLeft = MONO ? prm.nItems : prm.nItemsLeft, nSide = side ? prm.nItems - nLeft : nLeft;
    const MapView view{prm.items + (side ? nLeft : 0u), nSide, side ? 0u : nSide, side ? nLeft : 0u, prm.recs + side * prm.P, int(prm.P),
                       side ? 0 : int(prm.P), prm.mapped ? prm.mapped + (size_t(task) * (MONO ? 1 : 2) + side) * prm.P : nullptr,
                       MONO ? nullptr : sLate, int(prm.fixFrom[side])};
    const ChannelIndex at{N, side ? M : 0};
    float *win = lds + XFLOATS;
    // mono modes: the kSpecBins csf entries the reference leaves complex (complex_dc.hpp), behind the winners; written during the
    // recombination (nothing else uses the area), read by the pixels of prm.lowPixels after the mapping
    float *spec = win + (prm.nItems > 72u ? prm.nItems : 72u);
    MapPixelsBalanced<5, T, ChannelIndex> mapper;
    StftParams sp{};
    sp.weights = prm.weights; sp.invSize = prm.invSize;
#ifdef SGZ_DEBUG
    sp.phaseClock = prm.phaseClock; sp.ablate = prm.clkUnit << 16;
#endif

    RCLK(0);
    v2 c[R];
    {
        // ---------------------------------------------------------------- load + window: z[n] = (x[2n] w[2n], x[2n+1] w[2n+1])
        // 8-byte loads, R of them per thread, in batches of B with the next batch in flight while this one is multiplied (the
        // scheduling barriers keep instruction selection from hoisting all 2R loads to the top: 128 registers of raw samples)
        // which input channels feed this workgroup's signal: Separate: channel `side`; MidSide: (l + r) / 2 on side 0, (l - r) / 2 on side 1
        // (prepareTransform's MidSide case, then the same split as Separate); mono: l, r, (l + r) / 2 or (l - r) / 2
        const int firstCh = MONO ? (prm.mode == SGZ_CH_RIGHT ? 1 : 0) : (mixed ? 0 : side);
        const float *X = prm.planar + size_t(2 * pair + firstCh) * prm.chStride + size_t(frame) * prm.hop;
        // all R sample pairs are requested at once (64 registers) and multiplied by the window in place
        // (element = compile-time part + tid.  Pinning the compile-time part to scalar base registers -- `global_load v, v_lane, s[base]`, no
        // vector address arithmetic per load -- was measured on one box against this form: cfg2 -1 %, cfg5 +1.3 %: not kept)
        auto elemOf = [&](int e) { const int u = e / R1, j = e % R1; return T * u + RR * j; };
        const uint32_t lane8 = uint32_t(tid) * 8u;
#pragma unroll
        for (int i = 0; i < R; ++i) { const float2 xv = ldg(reinterpret_cast<const float2 *>(X) + elemOf(i), lane8); c[i] = v2{xv.x, xv.y}; }
        if (mixed) {
            // (l +- r) w 0.5 (prepareTransform, TransformDSP.inl:92-135): the right channel comes in batches of 8 pairs on top of the left
            const float *Y = X + prm.chStride;
            const float sgn = (MONO ? prm.mode == SGZ_CH_SIDE : side == 1) ? -1.f : 1.f;
            constexpr int YB = LR1 == 5 ? 4 : 8;        // 1024 threads: 128 registers, 64 of them hold the left channel
#pragma unroll
            for (int b0 = 0; b0 < R; b0 += YB) {
                float2 y[YB];
#pragma unroll
                for (int i = 0; i < YB; ++i) y[i] = ldg(reinterpret_cast<const float2 *>(Y) + elemOf(b0 + i), lane8);
#pragma unroll
                for (int i = 0; i < YB; ++i) c[b0 + i] = v2{c[b0 + i].x + sgn * y[i].x, c[b0 + i].y + sgn * y[i].y};
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (WCOS) {
            // w[n] = p0 + p1 cos(theta_n), theta_n = 2 pi n / N, n = 2 (col + R^2 j) + e: theta = phi(col, e) + 2 pi j / R1 -- the phase of the
            // column's first pair comes from a 16 KB table, the step to the next pair is a rotation by a compile-time angle
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 ph = ldg(prm.winPhase + T * u, uint32_t(tid) * 16u);
#pragma unroll
                for (int j = 0; j < R1; ++j) {
                    constexpr int S32 = 32 / R1;
                    const float cj = cos32((j * S32) % 32 <= 16 ? (j * S32) % 32 : 32 - (j * S32) % 32);
                    const float sj = (j * S32) % 32 <= 16 ? sin32((j * S32) % 32) : -sin32(32 - (j * S32) % 32);
                    const float ce = ph.x * cj - ph.y * sj, co = ph.z * cj - ph.w * sj;
                    const int i = u * R1 + j;
                    c[i] = v2{c[i].x * (prm.winP0 + prm.winP1 * ce), c[i].y * (prm.winP0 + prm.winP1 * co)};
                }
            }
        } else {
            // the window in batches of B pairs, two batches in flight
            constexpr int B = LR1 >= 4 ? 4 : 8;
            float2 wa[B], wb[B];
#pragma unroll
            for (int i = 0; i < B; ++i) wa[i] = ldg(reinterpret_cast<const float2 *>(prm.window) + elemOf(i), lane8);
#pragma unroll
            for (int b0 = 0; b0 < R; b0 += 2 * B) {
#pragma unroll
                for (int i = 0; i < B; ++i) wb[i] = ldg(reinterpret_cast<const float2 *>(prm.window) + elemOf(b0 + B + i), lane8);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < B; ++i) { c[b0 + i] = v2{c[b0 + i].x * wa[i].x, c[b0 + i].y * wa[i].y}; asm volatile("" : "+v"(c[b0 + i])); }
                if (b0 + 2 * B < R) {
#pragma unroll
                    for (int i = 0; i < B; ++i) wa[i] = ldg(reinterpret_cast<const float2 *>(prm.window) + elemOf(b0 + 2 * B + i), lane8);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < B; ++i) { c[b0 + B + i] = v2{c[b0 + B + i].x * wb[i].x, c[b0 + B + i].y * wb[i].y}; asm volatile("" : "+v"(c[b0 + B + i])); }
            }
        }
    }
    if (mixed) {
#pragma unroll
        for (int i = 0; i < R; ++i) c[i] = v2{c[i].x * 0.5f, c[i].y * 0.5f};
    }
    __builtin_amdgcn_sched_barrier(0);
    RCLK(13);
    // -------------------------------------------------------------------------- pass 1: radix R1 per column, times W_M^{c q1}
    pass1Columns<R, R1, U>(c);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        constexpr int NB = R1 / 4 - 1;
        float2 a[3], b[NB];
        const int col = tid + T * u;
        // W^c and W^{4c} from the table, the other rows as their powers (two loads per column instead of 3 + NB: the table is 8 KB per
        // row and workgroup, and what a workgroup fetches costs as much as what it computes)
        auto sq = [](float2 w) { return float2{w.x * w.x - w.y * w.y, 2.f * w.x * w.y}; };
        auto mul = [](float2 p, float2 q) { return float2{p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x}; };
        a[0] = ldg(prm.tw1 + (T * u), uint32_t(tid) * 8u);
        if (NB > 0) b[0] = ldg(prm.tw1 + (T * u + 3 * RR), uint32_t(tid) * 8u);
        a[1] = sq(a[0]); a[2] = mul(a[1], a[0]);
#pragma unroll
        for (int i = 1; i < NB; ++i) b[i] = (i & 1) ? sq(b[i / 2]) : mul(b[i - 1], b[0]);
#pragma unroll
        for (int q = 1; q < R1; ++q) {
            const int qa = q >> 2, qb = q & 3;
            v2 w;
            if (qa == 0) w = v2{a[qb - 1].x, a[qb - 1].y};
            else if (qb == 0) w = v2{b[qa - 1].x, b[qa - 1].y};
            else w = cmul(v2{b[qa - 1].x, b[qa - 1].y}, v2{a[qb - 1].x, a[qb - 1].y});
            const int i = u * R1 + brev(q, LR1);
            c[i] = cmul(c[i], w);
        }
    }
    RCLK(1);
    // -------------------------------------------------------------------------- exchange 1: two rounds of R1 x 512 complex values
    {
        v2 *lds2 = reinterpret_cast<v2 *>(lds);
        auto writeRound = [&](int r) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int col = tid + T * u;
                if ((col >> 9) == r) {
#pragma unroll
                    for (int q = 0; q < R1; ++q) lds2[q * 512 + (col & 511)] = c[u * R1 + brev(q, LR1)];
                }
            }
        };
        const int rd = q1 * 512 + ix;
        v2 lo[R / 2];
        writeRound(0);
        ldsBarrier();
#pragma unroll
        for (int h = 0; h < R / 2; ++h) lo[h] = lds2[rd + R * h];           // c_hi = h
        ldsBarrier();
        writeRound(1);
        ldsBarrier();
#pragma unroll
        for (int h = 0; h < R / 2; ++h) {
            c[h + R / 2] = lds2[rd + R * h];                                // c_hi = 16 + h
            c[h] = lo[h];
        }
    }
    ldsBarrier();                                                        // every wave has read exchange 1: the tiles may overwrite it
    RCLK(2);
    // -------------------------------------------------------------------------- pass 2 (c_lo = ix): radix R over c_hi
    difPacked<R, R, 0>(c);
    {
        TwFactors<LR> tw;
        tw.load(prm.tw2, ix, R);
        tw.apply(c);
    }
    RCLK(3);
    // -------------------------------------------------------------------------- exchange 2: wave-local R x R transposes
    {
        const int tile = group * TILE;
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
    RCLK(4);
    // -------------------------------------------------------------------------- pass 3 (q2 = ix): radix R over c_lo
    const float2 wk = ldg(prm.twPost, uint32_t(q1 + R1 * ix) * 8u);           // W_N^{kc}, for the recombination (a bin's is W_N^{kc} W_{2R}^{m3})
    difPacked<R, R, 0>(c);
    RCLK(5);
    // Z[kc + T m3] at register brev(m3), kc = q1 + R1 ix
    const int kc = q1 + R1 * ix;
    if (tid == 0) {                                                         // column 0 (k = T m3) pairs registers inside thread 0
#pragma unroll
        for (int m3 = 0; m3 < R; ++m3) {
            lds[SCRATCH + 2 * m3] = c[brev(m3, LR)].x;
            lds[SCRATCH + 2 * m3 + 1] = c[brev(m3, LR)].y;
        }
        // this channel's Nyquist bin X[M] = Re Z[0] - Im Z[0]: csf[N/2] is settled by whichever channel finishes second (below)
        sNyOwn = c[0].x - c[0].y;
        if (!MONO) __hip_atomic_store(prm.ny + self, sNyOwn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Column 0 (k = T m3, all in thread 0) pairs m3 with R - m3 inside one thread and holds DC / Nyquist: lanes 0 .. R/2 of wave 0 redo it
    // from thread 0's scratch copy right away (the scratch is not part of the tiles), keep the values and store them after the
    // workgroup's own stores.
    float fixA = 0.f, fixB = 0.f;
    if (tid <= R / 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (tid >= 1) {
            const int m3 = tid;                                             // k1 = T m3 and its mirror k2 = T (R - m3)  (m3 = R/2: one bin)
            const v2 a = v2{lds[SCRATCH + 2 * m3], lds[SCRATCH + 2 * m3 + 1]};
            const v2 b = v2{lds[SCRATCH + 2 * (R - m3)], lds[SCRATCH + 2 * (R - m3) + 1]};
            const float cs = cospif(float(m3) * (1.0f / 32.0f)), sn = sinpif(float(m3) * (1.0f / 32.0f));   // W_N^{T m3} = W_{2R}^{m3} = (cos, -sin)
            fixA = realBinMag(a, b, v2{cs, -sn});
            fixB = realBinMag(b, a, v2{-cs, -sn});                          // W_{2R}^{R - m3} = (-cos, -sin)
        } else {
            fixA = 0.5f * (lds[SCRATCH] + lds[SCRATCH + 1]);                // csf[0] = Re(csf[0]) * 0.5 / csf[N] = Im(csf[0]) * 0.5 (:861-862): X_c[0] / 2, signed
            if (MONO) { fixA = __builtin_fabsf(fixA); fixB = 0.5f * (lds[SCRATCH] - lds[SCRATCH + 1]); }   // |X[0]| / 2 and X[N/2] / 2 (:547-552)
        }
    }
    // ---- recombination.  a = Z[k] (own register m3 < R/2), b = Z[M - k] (lane L ^ R, register R-1-m3):
    //   2 E = a + conj b,  2 W O = -i w (a - conj b):   2 X[k] = 2E + 2WO   and   2 X[M - k] = conj(2E - 2WO)
    // so ONE evaluation gives the magnitudes of both bins of the pair.  A lane does this for its registers m3 < R/2; their mirrors are
    // the partner lane's registers >= R/2, and the partner does the same for ITS lower registers, whose mirrors are this lane's upper
    // ones: every bin of the lane pair is produced exactly once, with half the permutes, twiddles and adds of bin-by-bin evaluation.
    float magA[R / 2], magB[R / 2];
    {
        // lane holding Z[M - k]: L ^ R, except in slot 0 (q1 = 0: q2' = R - q2 ; q1 = R1/2: q2' = R-1-q2, same half)
        const int lane = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
        int plane = lane ^ R;
        if (slot == 0) plane = (lane & ~(R - 1)) | (half ? R - 1 - l : ((R - l) & (R - 1)));
        plane <<= 2;
        auto partnerOf = [&](float v) {
            return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plane, __builtin_bit_cast(int, v)));
        };
#pragma unroll
        for (int m3 = 0; m3 < R / 2; ++m3) {
            const int i = brev(m3, LR), ip = brev(R - 1 - m3, LR);
            const v2 a = c[i];
            const v2 b = v2{partnerOf(c[ip].x), partnerOf(c[ip].y)};
            const v2 w = m3 == 0 ? v2{wk.x, wk.y} : cmulConjK(v2{wk.x, wk.y}, v2{cos64(m3), sin64(m3)});
            const float ex = a.x + b.x, ey = a.y - b.y, dx = a.x - b.x, dy = a.y + b.y;
            const float ox = w.x * dy + w.y * dx, oy = w.y * dy - w.x * dx;      // -i w (dx + i dy)
            const float pr = ex + ox, pi = ey + oy, mr = ex - ox, mi = ey - oy;
            magA[m3] = 0.5f * __builtin_amdgcn_sqrtf(pr * pr + pi * pi);
            magB[m3] = 0.5f * __builtin_amdgcn_sqrtf(mr * mr + mi * mi);
            if (MONO && m3 == 0 && prm.lowCount[0] && kc >= 1 && kc <= 8) {
                // 2 X[kc] = (pr, pi), 2 X[M - kc] = (mr, -mi):  csf[N - kc] = Z[N - kc] = conj X[kc] (slot 8 - kc),
                // csf[N/2 + kc] = conj X[M - kc] (slot 8 + kc; kc = 8 has none)
                spec[2 * (8 - kc)] = 0.5f * pr; spec[2 * (8 - kc) + 1] = -0.5f * pi;
                if (kc < 8) { spec[2 * (8 + kc)] = 0.5f * mr; spec[2 * (8 + kc) + 1] = 0.5f * mi; }
            }
        }
    }
    RCLK(6);
    // csf[N/2 - 1] *= 0.5 (quirk Q3, TransformDSP.inl:864): the left channel's bin M - 1 = the mirror of bin 1
    if (!MONO && side == 0 && q1 == 1 && ix == 0) magB[0] *= 0.5f;
    if (nSide) mapper.prefetchTables(view, tid);
    ldsBarrier();                                                        // the tiles are dead: |X| may overwrite them
    {
        // left: bin k at position k; right: at position M - k (csf[N - k] = |X_R[k]|: csf order is ascending in LDS on both sides)
        // (two base addresses and compile-time offsets: a run-time stride costs a 64-bit multiply-add per store)
        const int up = kc + (kc >> 5), down = (M - kc) + ((M - kc) >> 5);
        int lowest = down - (R / 2 - 1) * PADSTRIDE;
        asm volatile("" : "+v"(lowest));                                    // (opaque: or the offsets are folded back into subtractions from `down`)
        float *pu = lds + up, *pd = lds + lowest;
        if (side == 0) {
#pragma unroll
            for (int m3 = 0; m3 < R / 2; ++m3) { pu[m3 * PADSTRIDE] = magA[m3]; pd[(R / 2 - 1 - m3) * PADSTRIDE] = magB[m3]; }
        } else {
#pragma unroll
            for (int m3 = 0; m3 < R / 2; ++m3) { pd[(R / 2 - 1 - m3) * PADSTRIDE] = magA[m3]; pu[m3 * PADSTRIDE] = magB[m3]; }
        }
    }
    if (tid <= R / 2) {                                                     // column 0, after this wave's own stores (one wave's LDS operations execute in order)
        auto put = [&](int k, float v) { const int i = side ? M - k : k; lds[i + (i >> 5)] = v; };
        if (tid >= 1) {
            put(T * tid, fixA);
            if (tid != R / 2) put(T * (R - tid), fixB);
        } else {
            put(0, fixA);
            if (MONO) { spec[16] = fixB; spec[17] = 0.f; }                   // csf[N/2] = Z[N/2] / 2 among the complex entries (slot 8)
            put(M, MONO ? fixB : 0.f);                                      // pairs: csf[N/2] is settled late, 0 can never win meanwhile (strict >)
        }
    }
    if (nSide && prm.mapped) mapper.prefetchWeights(sp, tid, nSide);
    ldsBarrier();
    RCLK(7);
    if (prm.binsOut) {                                                      // test hook: this side's half of csf, csf order
        // (csf[N/2] -- the left side's last entry, the right side's first -- is written by the settling workgroup alone)
        float *dst = prm.binsOut + size_t(task) * (N + 1) + (side ? M : 0);
        for (int i = tid; i <= M; i += T)
            if (MONO || i != (side ? 0 : M)) dst[i] = lds[i + (i >> 5)];
    }
    // ---- The pair exchange.  csf[N/2] = | X_L[M] + i X_R[M] | / 2 (TransformDSP.inl:863) with the pixels whose arg-max run ends on it (the
    // last offset of either side's scan, compared with a strict >), and the pixels whose tap window reaches over bin 0, need BOTH channels.
    // Nobody waits and nobody maintains caches: the exchanged arrays are fine-grained (coherent across the XCDs' L2s), accessed with
    // relaxed agent-scope atomics only, and ordered by the hardware's completion counters (a sequentially consistent agent-scope store /
    // load pair writes back and invalidates the XCD's whole L2: it cost 19 % of this kernel at N = 65536, 23 % at N = 32768).
    //   early  (here, before the mapping): wave 0 publishes this channel's Nyquist bin and lowest bins and, once those stores are
    //          acknowledged, raises flag1.
    //   end    if the partner's flag1 is up -- the usual case: the two workgroups run side by side -- this workgroup settles ITS OWN late
    //          pixels from the partner's published bins and is done.
    //          If not, it publishes the state of its late pixels, raises flag2 ("mine are unsettled"), waits for that store's
    //          acknowledgement and looks at flag1 once more: up now -> it settles itself after all; still down -> the partner, whose
    //          flag1 store then completes after this look, will find flag2 up at its own end (it waits for its flag1's acknowledgement
    //          before looking) and settles this side too.  Both may do it: identical values.  Late pixels have no other writer.
    if (!MONO && tid < 64) {
        if (prm.lowCount[0] + prm.lowCount[1]) {
            if (tid < kLowBins) {
                const int k = side ? N - tid : tid;
                __hip_atomic_store(prm.low + size_t(self) * kLowBins + tid, lds[at(k)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // wave 0 wrote ny and the low bins: acknowledged
        if (tid == 0) __hip_atomic_store(prm.nyFlag + 2 * self, prm.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (prm.mapped && nSide) mapper.run(sp, view, at, lds, win, tid, unit);
    RCLK(9);
    if (MONO) {
        // the pixels whose tap windows leave the magnitudes (wrap below bin 0: csf[N - j] = conj X[j], csf[N] = 0; or reach csf[N/2 ..]):
        // complex sums in the reference's order (complex_dc.hpp), from entries written before the last barrier; the mapping skipped them
        if (prm.mapped && prm.lowCount[0]) {
            float *out = prm.mapped + size_t(task) * prm.P;
            for (uint32_t i = tid; i < prm.lowCount[0]; i += T) {
                const uint32_t x = prm.lowPixels[i];
                out[x] = complexDcPixel(prm.recsFull[x], prm.weights, prm.invSize, N, prm.mode,
                                        [&](int k) { return k >= N ? 0.f : lds[k + (k >> 5)]; },
                                        [&](int sl) { return make_float2(spec[2 * sl], spec[2 * sl + 1]); });
            }
        }
        return;
    }
    __shared__ int sHave, sPartnerGaveUp;
    __syncthreads();                                                        // sLate is complete
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // flag1's store is at the coherence point (see above)
        sHave = __hip_atomic_load(prm.nyFlag + 2 * partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == prm.epoch ? 1 : 0;
        if (prm.pairTest == 2u && side == 1) {                              // (test hook: the left channel, dispatched first, is giving up)
            while (__hip_atomic_load(prm.nyFlag + 2 * partner + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != prm.epoch) __builtin_amdgcn_s_sleep(8);
            sHave = 1;
        }
        sPartnerGaveUp = __hip_atomic_load(prm.nyFlag + 2 * partner + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == prm.epoch ? 1 : 0;
        if (prm.pairTest == 1u || (prm.pairTest == 2u && side == 0)) sHave = 0;
    }
    __syncthreads();
    if (!sHave) {
        // the partner has not published yet: leave a note and look again
        if (tid < 128) __hip_atomic_store(prm.nyBest + size_t(self) * 128 + tid, sLate[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(prm.nyFlag + 2 * self + 1, prm.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sHave = __hip_atomic_load(prm.nyFlag + 2 * partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == prm.epoch ? 1 : 0;
            sPartnerGaveUp = 0;                                              // (a partner that has not even published cannot have given up)
            if (prm.pairTest == 2u) sHave = 0;
        }
        __syncthreads();
        if (!sHave) return;                                                  // the partner settles both sides at its end
    }
    // settle: s = 0 this side (own late state in LDS, own low bins in LDS), s = 1 the partner's side if it gave up (its state from the
    // published arrays).  Threads [0, 64): the pixels whose run ends on csf[N/2]; [128, 256): the pixels that reach over bin 0.
    {
#pragma clang fp contract(off)
        const float nyP = __hip_atomic_load(prm.ny + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float nyRe = side ? nyP : sNyOwn, nyIm = side ? sNyOwn : nyP;     // left channel's is the real part
        const float vM = 0.5f * __builtin_amdgcn_sqrtf(nyRe * nyRe + nyIm * nyIm);
        const float sqM = vM * vM + 0.f;                                    // Math::square(csf[offset]) with imag == 0
        if (tid == 0 && prm.binsOut) prm.binsOut[size_t(task) * (N + 1) + M] = vM;
        const float *lowP = prm.low + size_t(partner) * kLowBins;
        for (int who = 0; who < (sPartnerGaveUp ? 2 : 1); ++who) {
            const int s = who ? (side ^ 1) : side;                           // the side being settled
            const long u = (task << 1) | s;
            if (prm.mapped && tid < 64) {
                const uint32_t x = prm.fixFrom[s] + uint32_t(tid);
                if (x < prm.P) {
                    float b, own;
                    if (!who) { b = sLate[tid]; own = sLate[64 + tid]; }
                    else {
                        b = __hip_atomic_load(prm.nyBest + size_t(u) * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        own = __hip_atomic_load(prm.nyBest + size_t(u) * 128 + 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    prm.mapped[size_t(u) * prm.P + x] = sqM > b ? finishPixel<5>(prm.invSize * vM) : own;
                }
            }
            if (prm.mapped && tid >= 128 && tid < 256) {
                const uint32_t n = uint32_t(tid - 128);
                if (n < prm.lowCount[s]) {
                    const uint32_t x = prm.lowPixels[(s ? prm.lowCount[0] : 0u) + n];
                    const PixelRec rec = prm.recsFull[size_t(s) * prm.P + x];
                    float v[kMaxTaps], w[kMaxTaps];
                    int k = rec.a;
#pragma unroll
                    for (int i = 0; i < kMaxTaps; ++i) {                      // independent loads first
                        const bool on = i < rec.b;
                        // csf[k]: k < kLowBins is the left channel's bin k, k > N - kLowBins the right channel's bin N - k; this
                        // workgroup's own bins are in its LDS, the partner's in its published array
                        const bool left = k < kLowBins;
                        const int j = left ? k : N - k;
                        const bool mine = (side == 0) == left;
                        float val = 0.f;
                        if (on) val = mine ? lds[at(k)] : __hip_atomic_load(lowP + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v[i] = val;
                        w[i] = on ? prm.weights[rec.c + i] : 0.f;
                        if (on) k = (k == N) ? 0 : k + 1;
                    }
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < kMaxTaps; ++i)
                        if (i < rec.b) acc = acc + v[i] * w[i];               // taps in order (lanczosFilter restatement)
                    prm.mapped[size_t(u) * prm.P + x] = finishPixel<5>(prm.invSize * acc);
                }
            }
        }
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember what has been granted
static hipError_t grantLds(const void *kernel, size_t need, size_t (&granted)[64])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && granted[dev] >= need) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(need));
    if (e == hipSuccess && dev >= 0 && dev < 64) granted[dev] = need;
    return e;
}

hipError_t launchStftReal(const RealParams &prm, uint32_t N, hipStream_t stream)
{
    const bool mono = prm.mode != SGZ_CH_SEPARATE && prm.mode != SGZ_CH_MIDSIDE;
    const long units = prm.frames * long(prm.C) * (mono ? 1 : 2);
    if (units <= 0) return hipSuccess;
    const uint32_t M = N / 2;
    const size_t xFloats = size_t(((M + 1) + ((M + 1) >> 5) + 2) & ~1u);
    const uint32_t maxSide = mono ? prm.nItems : std::max(prm.nItemsLeft, prm.nItems - prm.nItemsLeft);
    const size_t ldsBytes = xFloats * 4 + size_t(std::max(maxSide, 72u)) * 4 + (mono ? 2 * kSpecBins * 4 : 0);
    static size_t granted[24][64] = {};
    const bool wcos = prm.winPhase != nullptr;
    auto go = [&](auto kern, int slot, unsigned threads, size_t limit) -> hipError_t {
        if (ldsBytes > limit) return hipErrorInvalidValue;
        if (hipError_t e = grantLds(reinterpret_cast<const void *>(kern), ldsBytes, granted[slot]); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(unsigned(units)), dim3(threads), ldsBytes, stream, prm);
        return hipSuccess;
    };
    hipError_t e;
    if (mono && (prm.mode == SGZ_CH_MERGE || prm.mode == SGZ_CH_SIDE)) {
        if (N == 32768) e = wcos ? go(&stftRealKernel<4, true, 3>, 18, 512, 80 * 1024) : go(&stftRealKernel<4, false, 3>, 19, 512, 80 * 1024);
        else if (N == 16384) e = wcos ? go(&stftRealKernel<3, true, 3>, 20, 256, 40 * 1024) : go(&stftRealKernel<3, false, 3>, 21, 256, 40 * 1024);
        else if (N == 65536) e = wcos ? go(&stftRealKernel<5, true, 3>, 22, 1024, 160 * 1024) : go(&stftRealKernel<5, false, 3>, 23, 1024, 160 * 1024);
        else return hipErrorNotSupported;
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    if (mono) {
        if (N == 32768) e = wcos ? go(&stftRealKernel<4, true, 1>, 6, 512, 80 * 1024) : go(&stftRealKernel<4, false, 1>, 7, 512, 80 * 1024);
        else if (N == 16384) e = wcos ? go(&stftRealKernel<3, true, 1>, 8, 256, 40 * 1024) : go(&stftRealKernel<3, false, 1>, 9, 256, 40 * 1024);
        else if (N == 65536) e = wcos ? go(&stftRealKernel<5, true, 1>, 10, 1024, 160 * 1024) : go(&stftRealKernel<5, false, 1>, 11, 1024, 160 * 1024);
        else return hipErrorNotSupported;
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    if (prm.mode == SGZ_CH_MIDSIDE) {
        if (N == 32768) e = wcos ? go(&stftRealKernel<4, true, 2>, 12, 512, 80 * 1024) : go(&stftRealKernel<4, false, 2>, 13, 512, 80 * 1024);
        else if (N == 16384) e = wcos ? go(&stftRealKernel<3, true, 2>, 14, 256, 40 * 1024) : go(&stftRealKernel<3, false, 2>, 15, 256, 40 * 1024);
        else if (N == 65536) e = wcos ? go(&stftRealKernel<5, true, 2>, 16, 1024, 160 * 1024) : go(&stftRealKernel<5, false, 2>, 17, 1024, 160 * 1024);
        else return hipErrorNotSupported;
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    if (N == 32768) e = wcos ? go(&stftRealKernel<4, true>, 0, 512, 80 * 1024) : go(&stftRealKernel<4, false>, 1, 512, 80 * 1024);
    else if (N == 16384) e = wcos ? go(&stftRealKernel<3, true>, 4, 256, 40 * 1024) : go(&stftRealKernel<3, false>, 5, 256, 40 * 1024);
    else if (N == 65536) e = wcos ? go(&stftRealKernel<5, true>, 2, 1024, 160 * 1024) : go(&stftRealKernel<5, false>, 3, 1024, 160 * 1024);
    else return hipErrorNotSupported;
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace sgz
