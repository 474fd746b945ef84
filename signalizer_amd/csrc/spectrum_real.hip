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
// it: a workgroup maps with 0 in its place and leaves its own Nyquist bin and the winning squares of the (top) pixels whose run ends
// there in HBM; realLateKernel, behind the launch, settles those pixels for both sides.
// Everything else the path needs (mono modes, Complex, Phase, zero-padded windows, views whose filter taps wrap around csf) stays
// on the whole-frame kernels; plan.cpp decides (Plan::realSplit).
#include <algorithm>

#include "real_common.hpp"

#ifdef SGZ_DEBUG
#define RCLK(slot)                                                                                                     \
    do {                                                                                                               \
        if (prm.phaseClock && (tid & 63) == 0 && unit == long(prm.clkUnit)) prm.phaseClock[16 * (tid >> 6) + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
// clkUnit == 0xffff: instead, every workgroup leaves (start, end) in the 100 MHz wall clock all CUs share, HW_ID and XCC_ID behind
// the 256 phase slots: the launch's schedule (tools/unit_trace.py)
#define RTRACE(which)                                                                                                  \
    do {                                                                                                               \
        if (prm.phaseClock && prm.clkUnit == 0xffffu && tid == 0) {                                                    \
            unsigned long long *t = prm.phaseClock + 256 + 4 * unit;                                                   \
            t[which] = __builtin_amdgcn_s_memrealtime();                                                               \
            if (which == 0) { t[2] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); t[3] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); } \
        }                                                                                                              \
    } while (0)
#else
#define RCLK(slot) do { } while (0)
#define RTRACE(which) do { } while (0)
#endif

namespace sgz {

// exchange 1's stores of one column as single ds_write_b64 (fft_common.hpp ldsWrite64)
template <int R1, int LR1, int BASE, int Q = 0>
__device__ __forceinline__ void e1WriteRun(uint32_t wa, const v2 (&c)[32])
{
    if constexpr (Q < R1) { ldsWrite64<(Q & 15) * 512 * 8>(wa + uint32_t(Q >> 4) * 65536u, c[BASE + brev(Q, LR1)]); e1WriteRun<R1, LR1, BASE, Q + 1>(wa, c); }
}

// ... of two neighbouring columns (registers [0, R1) and [R1, 2 R1)) as one ds_write_b128 per row
template <int R1, int LR1, int Q = 0>
__device__ __forceinline__ void e1WritePairRun(uint32_t wa, const v2 (&c)[32])
{
    if constexpr (Q < R1) { ldsWrite128<(Q & 15) * 512 * 8>(wa + uint32_t(Q >> 4) * 65536u, c[brev(Q, LR1)], c[R1 + brev(Q, LR1)]); e1WritePairRun<R1, LR1, Q + 1>(wa, c); }
}

// the R1-point DIFs of a thread's U = R / R1 columns (registers [u R1, (u + 1) R1))
template <int R, int R1, int U, int LR1, int u = 0>
__device__ __forceinline__ void pass1Columns(v2 (&c)[R])
{
    if constexpr (u < U) {
        ditPacked<LR1, u * R1>(c);
        pass1Columns<R, R1, U, LR1, u + 1>(c);
    }
}

// ---- "fetch diet" (round 6): what a thread's SECOND column needs from the tables is one fixed rotation away from the first column's.
// A workgroup fetches ~60 KB of tables beside its 128 KB of samples, the same bytes for every workgroup, through a per-CU fetch path that
// is as busy as the vector ALUs (NOTES.md, round 5): columns 2 tid and 2 tid + 1 differ by ONE sample pair, so the second column's window
// phase is the first's turned by a = 2 pi 2 / N, its pass-1 twiddle rows W_M^{c + 1} = W_M^c W_M^1 and W_M^{4 (c + 1)} = W_M^{4c} W_M^4.
// Rotations by tiny angles in the form x + (x (cos a - 1) - y sin a): cos a - 1 is a full-precision float (cos a itself would round to
// 1 - 2^-24), so the turned value is within an ulp of its magnitude.  32 bytes and three requests per thread less, eight packed
// operations more.  N = 32768 (M = 16384): a = 2 pi / 16384 for the phase and for W_M^1, 2 pi / 4096 for W_M^4.
// MEASURED (round 6, tools/ab.sh + tools/ab_rot.sh on one box, all 131 spectrum / golden / fuzz tests green with it): 16 KB of table
// bytes (of ~60) and 3 of a thread's ~29 requests less move nothing -- cfg2 24.6 | 24.6 us with the input L2-resident, 26.8 | 27.1 us
// from HBM, tail-free 155.9 | 156.2, cfg5 +-0.  The tables are not what the fetch path is busy with (they are the same lines for every
// workgroup of a CU and hit its L1); kept behind -DSGZ_DIET as the record of the experiment.
constexpr float kRotA1Cm1 = -0x1.3bd3ccp-24f, kRotA1Sin = 0x1.921fb4p-12f;      // cos(2 pi / 16384) - 1, sin(2 pi / 16384)
constexpr float kRotA4Cm1 = -0x1.3bd3c8p-20f, kRotA4Sin = 0x1.921faap-10f;      // cos(2 pi / 4096) - 1,  sin(2 pi / 4096)
// (cos t, cos t', sin t, sin t') -> the same of t + a, t' + a
__device__ __forceinline__ float4 rotatePhaseForward(const float4 ph, const float cm1, const float sn)
{
    const v2 pc = v2{ph.x, ph.y}, ps = v2{ph.z, ph.w};
    const v2 kc = v2{cm1, cm1}, ks = v2{sn, sn};
    const v2 c2 = __builtin_elementwise_fma(pc, kc, __builtin_elementwise_fma(ps, -ks, pc));
    const v2 s2 = __builtin_elementwise_fma(ps, kc, __builtin_elementwise_fma(pc, ks, ps));
    return float4{c2.x, c2.y, s2.x, s2.y};
}
// w = (cos t, -sin t) -> (cos(t + a), -sin(t + a)):  w (cos a - i sin a)
__device__ __forceinline__ float2 rotateTwiddle(const float2 w, const float cm1, const float sn)
{
    const v2 v = v2{w.x, w.y}, sw = v2{w.y, w.x};
    const v2 r = __builtin_elementwise_fma(v, v2{cm1, cm1}, __builtin_elementwise_fma(sw, v2{sn, -sn}, v));   // (x + y sin a, y - x sin a) + (x, y)(cos a - 1)
    return float2{r.x, r.y};
}

// Everything behind the barrier that completes a side's magnitudes in LDS: the test hook that writes them out, the pair exchange,
// the pixel map (chunk_map.hpp) and the settlement of the pixels that need both channels.  Shared by the transform kernel and by
// realMapFromBinsKernel (sgz_stage_map_from_bins: the same code maps injected bins, so "the mapping is bit-exact given the bins"
// is tested on the very functions the bench kernel runs).
template <int LR1, int MIX, typename P>
__device__ __forceinline__ void realMapSettle(const P &prm, float *lds, const int tid, const int side, const long task, const long self,
                                              const ChunkTables &tb, ChunkMap<(1 << (LR1 + 5))> &mapper, float *re, float *ce, float *spec)
{
    constexpr int R = 32, R1 = 1 << LR1, T = R1 * R, M = R1 * R * R, N = 2 * M;
    constexpr bool MONO = MIX == 1 || MIX == 3;
    const ChannelIndex at{N, side ? M : 0};
    [[maybe_unused]] const long unit = MONO ? task : self;                   // (debug clocks)
    if (prm.binsOut) {                                                      // test hook: this side's half of csf, csf order
        // (csf[N/2] -- the left side's last entry, the right side's first -- is written by the settling workgroup alone)
        float *dst = prm.binsOut + size_t(task) * (N + 1) + (side ? M : 0);
        for (int i = tid; i <= M; i += T)
            if (MONO || i != (side ? 0 : M)) dst[i] = lds[chunkPos(i)];
    }
    // ---- What needs BOTH channels is left to realLateKernel (below), which runs behind this launch: csf[N/2] = |X_L[M] + i X_R[M]| / 2
    // (TransformDSP.inl:863) with the pixels whose arg-max run ends on it, and the pixels whose tap window reaches over bin 0.  This
    // workgroup only leaves what that kernel needs in HBM -- its Nyquist bin (ny, stored by the transform), its lowest bins, and (from the
    // map) the winning squares of its top pixels, whose values so far ignore csf[N/2] (0 in its place can never win: strict >).
    // (Until round 3 the two channel workgroups of a frame settled these pixels between themselves with flags in fine-grained memory:
    // three dependent memory round trips at the end of every workgroup: 1.5 us of a 37 us launch in an ablation build.)
    if (!MONO && (prm.lowCount[0] + prm.lowCount[1]) && tid < kLowBins) prm.low[size_t(self) * kLowBins + tid] = lds[at(side ? N - tid : tid)];
    RCLK(8);
    mapper.run(tb, at, lds, re, ce, prm.invSize, tid);
    RCLK(9);
    RTRACE(1);
    if (MONO) {
        // the pixels whose tap windows leave the magnitudes (wrap below bin 0: csf[N - j] = conj X[j], csf[N] = 0; or reach csf[N/2 ..]):
        // complex sums in the reference's order (complex_dc.hpp), from entries written before the last barrier; the mapping skipped them
        if (prm.mapped && prm.lowCount[0]) {
            float *out = prm.mapped + size_t(task) * prm.P;
            for (uint32_t i = tid; i < prm.lowCount[0]; i += T) {
                const uint32_t x = prm.lowPixels[i];
                out[x] = complexDcPixel(prm.recsFull[x], prm.weights, prm.invSize, N, prm.mode,
                                        [&](int k) { return k >= N ? 0.f : lds[chunkPos(k)]; },
                                        [&](int sl) { return make_float2(spec[2 * sl], spec[2 * sl + 1]); });
            }
        }
        return;
    }
}

// MONO: SpectrumChannels Left / Right / Merge / Side -- ONE real signal per (frame, pair) (the reference transforms it as a complex frame
// with a zero imaginary part, TransformDSP.inl:59-135): one workgroup per task, no pair exchange.  csf[0] = |X[0]| / 2 and
// csf[N/2] = X[N/2] / 2 (:547-552; the latter stays signed: the reference leaves it complex, and X[N/2] of a real signal is real).
template <int LR1, bool WCOS, int MIX = 0>                 // MIX: 0 Separate (two channel workgroups), 1 mono Left / Right, 2 MidSide (two workgroups on mid and side), 3 mono Merge / Side
__global__ void __launch_bounds__(1 << (LR1 + 5), 4) stftRealKernel(const RealParams launchPrm)
{
    constexpr int LR = 5, R = 32, R1 = 1 << LR1, T = R1 * R, RR = R * R, M = R1 * RR, N = 2 * M, U = R / R1;
    constexpr bool MONO = MIX == 1 || MIX == 3;
    constexpr bool mixed = MIX >= 2;                    // the signal is (l +- r) / 2: compile-time, the second channel's loads cost registers
    constexpr int PADSTRIDE = chunkPos(T);              // padded distance between k and k + T
    constexpr int ROW = R + 2, TILE = R * ROW;            // exchange 2: rows of 32 + 2 floats (8-byte aligned rows: the transposed reads are ds_read_b64, free of bank conflicts)
    constexpr int XFLOATS = realXFloats(M);             // this side's |X| array (padded) -- the map's tile and chunk maxima follow it
    static_assert((T / R) * TILE <= XFLOATS, "the exchange-2 tiles end below the twiddle table");
    constexpr int TAB = (XFLOATS + 3) & ~3;               // the pass-2 twiddle table (16-byte aligned)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid0 = threadIdx.x;

    // WALK (N = 65536, pairs, window evaluated in the kernel): 1024 threads x 128 registers are the CU's whole register file, so ONE workgroup
    // is resident and its phases run one after the other -- the 256 KB of samples arrive through the CU's ~11 B/clock fetch path while
    // nothing computes (tools/phase_clocks.py: 22 k of a workgroup's 57 k clocks).  The launch is therefore one workgroup per CU that WALKS
    // over the units a one-unit-per-workgroup launch would have given its slot (b, b + #workgroups, ...: the same XCD-aware order), and each
    // wave requests the first half of the NEXT unit's rows as soon as its own magnitudes are out of the registers those rows land in --
    // the requests are served while the slower waves still transform and the map runs; the second half follows at the top of the loop.
    constexpr bool WALK = LR1 == 5 && MIX == 0 && WCOS;
    const uint32_t totalUnits = WALK ? uint32_t(launchPrm.frames) * launchPrm.C * 2u : gridDim.x;
    uint32_t walkIndex = blockIdx.x;
    UnitId uid = unitOfIndex<MONO>(launchPrm, walkIndex, totalUnits);     // (frame, pair, channel) or (frame, pair); real_common.hpp
    [[maybe_unused]] bool firstUnit = true;
    // Wave priorities for a launch of two full dispatch generations and a partial third (cfg2: 696 workgroups on 256 CUs, two resident
    // per CU).  tools/unit_trace.py: workgroups b and b + #CUs share a CU, the third generation starts in the slots the first frees
    // and the launch ends when IT ends; its workgroups share their CU with second-generation ones that have ~10 us of slack.  Third
    // generation at priority 3, first at 2, second at 1: -0.6 us of a 28.7 us launch at the sustained clock (tools/ab3.sh; every
    // assignment with the last generation on top measures within 0.1 us of this one, 3 / 0 / 3 half the gain, 3 / 0 / 0 a loss).
    // Longer launches are left alone.  (A speed assumption only.)
    if constexpr (LR1 == 4) {
        const uint32_t cus = launchPrm.roundSize >> 1;
        // The two workgroups that share a CU from the first clock of a launch (b and b + #CUs) would run IN PHASE -- both fetching, both in
        // their LDS exchanges, both in the map's scalar-heavy scan at the same time -- and take 65 k clocks each where two workgroups half
        // a life apart take 41 k (tools/phase_clocks.py): the second one starts ~10 k clocks late.  cfg2 launch -2.3 ... -5.5 % with the
        // input L2-resident, -5 % from HBM (tools/ab_rot.sh, two boxes; 4 k clocks: half the gain, 16 k: a loss); the workgroups of a long launch
        // stay out of phase afterwards: 2784 workgroups 172.8 -> 156.6 us.  (A speed assumption only.)
#ifndef SGZ_STAGGER
#define SGZ_STAGGER 5
#endif
        if (cus && !launchPrm.pipelined && gridDim.x >= 2u * cus && blockIdx.x >= cus && blockIdx.x < 2u * cus) {       // (a launch with a FULL second generation: a partial one -- 1.5 generations -- loses 1.3 us to the delay)
#pragma unroll
            for (int k = 0; k < SGZ_STAGGER; ++k) __builtin_amdgcn_s_sleep(32);      // (32 x 64 clocks per step)
        }
        // (both are for a launch that has the chip to itself: with other launches beside it -- sgz_render_queue -- the delay and the priorities
        // cost 3.5 us per render, tools/pipeline_depth.py; RealParams::pipelined)
#ifndef SGZ_NO_PRIO
        if (cus && !launchPrm.pipelined && gridDim.x > 2u * cus && gridDim.x <= 3u * cus) {
#else
        if (false) {
#endif
            const uint32_t generation = blockIdx.x / cus;
            if (generation == 0u) __builtin_amdgcn_s_setprio(2); else if (generation == 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3);
        }
    }
    constexpr bool PAIRED = U == 2;                          // thread -> columns 2 tid, 2 tid + 1 (otherwise tid + T u): see the sample loads
    constexpr int COLSTEP = PAIRED ? 1 : T, COLLANE = PAIRED ? 2 : 1;     // column of (tid, u) = COLLANE tid + COLSTEP u
    constexpr bool FRONT = LR1 >= 4 && !mixed && WCOS;       // (the mixed modes and a fetched window hold more values in flight: they would spill)
#ifdef SGZ_DIET                                              // (measured +-0 with the input L2-resident, +0.2 ... 0.6 us from HBM: NOTES.md round 6 -- off by default)
    constexpr bool DIET = PAIRED && FRONT && LR1 == 4;       // the second column's phase and twiddle rows by rotation of the first's (above)
#else
    constexpr bool DIET = false;
#endif
    [[maybe_unused]] float4 phase[U];
    [[maybe_unused]] float2 twA[U], twB[U];
    [[maybe_unused]] float4 tw2piece, tw2tail;                  // (T = 512: threads 0 .. 31 carry a second piece of the 544)
    v2 c[R];
    for (;;) {
    // (WALK: the parameter block is read in place, through a pointer that is opaque once per unit: real_common.hpp WalkParams)
    typename ParamsOf<WALK>::T &prm = *ParamsOf<WALK>::get(launchPrm);
    // which input channels feed a unit's signal: Separate: channel `side`; MidSide: (l + r) / 2 on side 0, (l - r) / 2 on side 1
    // (prepareTransform's MidSide case, then the same split as Separate); mono: l, r, (l + r) / 2 or (l - r) / 2
    auto samplesOf = [&](const UnitId &u) {
        const int firstCh = MONO ? (prm.mode == SGZ_CH_RIGHT ? 1 : 0) : (mixed ? 0 : u.side);
        return prm.planar + size_t(2 * u.pair + firstCh) * prm.chStride + size_t(u.frame) * prm.hop;
    };
    auto elemOf = [&](int e) { const int u = e / R1, j = e % R1; return (PAIRED ? u : T * u) + RR * j; };
    // all R sample pairs are requested at once (64 registers) and multiplied by the window in place
    // (element = compile-time part + tid.  Pinning the compile-time part to scalar base registers -- `global_load v, v_lane, s[base]`, no
    // vector address arithmetic per load -- was measured on one box against this form: cfg2 -1 %, cfg5 +1.3 %: not kept)
    // PAIRED (two columns per thread, N = 32768): the thread owns the NEIGHBOURING columns 2 tid and 2 tid + 1, so that a sample request is
    // 16 bytes per lane (global_load_dwordx4: 1 KB per wave-instruction) -- the per-CU fetch path serves 8-byte requests at 0.54-0.70 of
    // its 16-byte rate (MI355X_MICROARCH.md), and the samples' arrival is the longest phase of a workgroup's life (tools/phase_clocks.py)
    // part 0: every row; WALK: part 1 = rows 0 .. 7 and 16 .. 23 (the window pairs row j with row j + 16), part 2 = the others
    auto requestSamples = [&](const float *X, const int tid, auto partTag) {
        constexpr int part = decltype(partTag)::value;
        const uint32_t lane8 = uint32_t(tid) * (PAIRED ? 16u : 8u);
        if constexpr (PAIRED) {
            typedef float v4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int j = 0; j < R1; ++j) { const v4 xv = ldgPinned<v4>(X, uint32_t(RR * j) * 8u, lane8); c[j] = v2{xv.x, xv.y}; c[R1 + j] = v2{xv.z, xv.w}; }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (part == 0 || ((i >> 3) & 1) == part - 1) c[i] = ldgPinned<v2>(X, uint32_t(elemOf(i)) * 8u, lane8);
        }
    };
    // Every table value this thread needs before the first barrier is REQUESTED behind the samples and in front of the first wait:
    // the window phases, the two fetched pass-1 twiddle rows of every column and this thread's piece of the pass-2 table (which goes to
    // LDS in front of exchange 1).  Requested where they are used -- as this kernel did until round 5 -- each is a memory round trip of
    // its own in the workgroup's dependent chain: table -> LDS copy (waited for before the first sample was requested), samples,
    // second column's phases, first column's twiddles, second column's twiddles.  (N = 16384: 256 threads x four columns at the
    // 128-register limit: requested where used, as before.)
    auto requestTables = [&](const int tid) {
        if constexpr (FRONT) {
            if constexpr (WALK) tw2piece = prm.tw2Full[tid < kTw2Floats / 4 ? tid : 0];     // (unconditional: a conditional one makes the registers loop-carried)
#pragma unroll
            for (int u = 0; u < (DIET ? 1 : U); ++u) {
                if (WCOS) phase[u] = ldg(prm.winPhase + COLSTEP * u, uint32_t(tid) * 16u * COLLANE);
                twA[u] = ldg(prm.tw1 + (COLSTEP * u), uint32_t(tid) * 8u * COLLANE);
                twB[u] = ldg(prm.tw1 + (COLSTEP * u + 3 * RR), uint32_t(tid) * 8u * COLLANE);
            }
            if constexpr (!WALK) { if (tid < kTw2Floats / 4) tw2piece = prm.tw2Full[tid]; }
            if (T < kTw2Floats / 4 && tid < kTw2Floats / 4 - T) tw2tail = prm.tw2Full[T + tid];
        }
    };
    if constexpr (WALK) {
        if (firstUnit) requestSamples(samplesOf(uid), tid0, std::integral_constant<int, 1>{});
        firstUnit = false;
    }
    // (WALK: the thread index is opaque per unit, or everything derived from it -- LDS addresses, lane offsets, twiddle indices -- is hoisted
    // out of the loop and held in registers through every phase: 33 spilled registers in the first build)
    const int tid = WALK ? opaque(tid0) : tid0;
    const int slot = tid >> 6, half = (tid >> 5) & 1, l = tid & 31, group = tid >> 5;
    const int q1 = half ? (slot == 0 ? R1 / 2 : R1 - slot) : slot;
    const int ix = half ? R - 1 - l : l;                // c_lo in pass 2, q2 in pass 3
    const uint32_t lane8 = uint32_t(tid) * (PAIRED ? 16u : 8u);
    [[maybe_unused]] const long unit = MONO ? long(uid.task) : long(uid.self);   // (debug clocks)
    const int side = uid.side, pair = uid.pair;
    const long task = uid.task, frame = uid.frame, self = uid.self;

    // map tables of this side (chunk_map.hpp)
    const uint32_t maxSlots = prm.chunkSlots[0] > prm.chunkSlots[1] ? prm.chunkSlots[0] : prm.chunkSlots[1];
    const ChunkTables tb{prm.chunkEnds + side * T, prm.chunkReBase + side * T,
                         reinterpret_cast<const uint2 *>(prm.chunkRec) + size_t(side) * prm.P, prm.recs + size_t(side) * prm.P, prm.weights12,
                         prm.mapped ? prm.mapped + (size_t(task) * (MONO ? 1 : 2) + side) * prm.P : nullptr,
                         MONO ? nullptr : prm.nyBest + size_t(self) * 64, int(prm.fixFrom[side]), int(prm.P), side != 0,
#ifdef SGZ_DEBUG
                         (prm.phaseClock && unit == long(prm.clkUnit)) ? prm.phaseClock : nullptr};
#else
                         nullptr};
#endif
    const ChannelIndex at{N, side ? M : 0};
    float *re = lds + XFLOATS, *ce = re + maxSlots + 1;
    // mono modes: the kSpecBins csf entries the reference leaves complex (complex_dc.hpp), behind the maxima; written during the
    // recombination (nothing else uses the area), read by the pixels of prm.lowPixels after the mapping
    // behind them (and behind the pass-2 twiddle table, which shares their place but is still being read by slow waves when fast ones
    // are past pass 3): column 0's 2R floats of scratch, then the mono modes' complex entries
    const int SCRATCH = XFLOATS + int(realExtraFloats(maxSlots, T, LR1 >= 4));
    float *spec = lds + SCRATCH + 2 * R;
    ChunkMap<T> mapper;

    RCLK(0);
    RTRACE(0);
    if constexpr (LR1 >= 4 && !FRONT) {
        // the pass-2 twiddle table -> LDS (8.5 KB behind the exchange areas; the map's maxima take the place later)
        for (int i = tid; i < kTw2Floats / 4; i += T) reinterpret_cast<float4 *>(lds + TAB)[i] = prm.tw2Full[i];
    }
    {
        // ---------------------------------------------------------------- load + window: z[n] = (x[2n] w[2n], x[2n+1] w[2n+1])
        const float *X = samplesOf(uid);
        if constexpr (WALK) {
            // part 1 was requested during the previous unit and is in place: the tables go FIRST, so that its rows are windowed while
            // part 2 arrives, and the pass-2 table's piece is parked in LDS as soon as it is there (four registers less through pass 1)
            requestTables(tid);
            __builtin_amdgcn_sched_barrier(0);
            requestSamples(X, tid, std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
            if (tid < kTw2Floats / 4) reinterpret_cast<float4 *>(lds + TAB)[tid] = tw2piece;
        } else {
            requestSamples(X, tid, std::integral_constant<int, 0>{});
            requestTables(tid);
        }
        if constexpr (FRONT) __builtin_amdgcn_sched_barrier(0);
        if (mixed) {
            // (l +- r) w 0.5 (prepareTransform, TransformDSP.inl:92-135): the right channel comes in batches of 8 pairs on top of the left
            const float *Y = X + prm.chStride;
            const float sgn = (MONO ? prm.mode == SGZ_CH_SIDE : side == 1) ? -1.f : 1.f;
            constexpr int YB = (LR1 == 5 || (PAIRED && !WCOS)) ? 4 : 8;        // 1024 threads: 128 registers, 64 of them hold the left channel
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
            // column's first pair comes from a 16 KB table, the step to the next pair is a rotation by a compile-time angle.  Evaluated on
            // (even, odd) sample pairs with packed operations, and only for j < R1 / 2: j + R1 / 2 is half a turn further, its cosine the
            // negative -- w[j] = p0 + t, w[j + R1/2] = p0 - t with t = p1 cos(theta_j) (p1 is folded into the table).
            const v2 p0 = v2{prm.winP0, prm.winP0};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 ph = (DIET && u == 1) ? rotatePhaseForward(phase[0], kRotA1Cm1, kRotA1Sin)
                                  : FRONT ? phase[u] : ldg(prm.winPhase + COLSTEP * u, uint32_t(tid) * 16u * COLLANE);     // p1 x (cos even, cos odd, sin even, sin odd) of the column's first pair
                const v2 pc = v2{ph.x, ph.y}, ps = v2{ph.z, ph.w};
#pragma unroll
                for (int j = 0; j < R1 / 2; ++j) {
                    constexpr int S32 = 32 / R1;
                    const int a32 = j * S32;                                    // angle in 32nds of a turn, < 16
                    const int i = u * R1 + j, i2 = i + R1 / 2;
                    if constexpr (LR1 == 3) {
                        // (the 256-thread kernel runs at the 128-register limit with four columns' phases in flight: scalar form, same symmetry)
                        const float cj = cos32(a32), sj = sin32(a32);
                        const float te = a32 == 0 ? ph.x : ph.x * cj - ph.z * sj, to = a32 == 0 ? ph.y : ph.y * cj - ph.w * sj;
                        c[i] = v2{c[i].x * (prm.winP0 + te), c[i].y * (prm.winP0 + to)};
                        c[i2] = v2{c[i2].x * (prm.winP0 - te), c[i2].y * (prm.winP0 - to)};
                        continue;
                    }
                    v2 t = pc;
                    if (a32 != 0) {
                        const v2 k = v2{cos32(a32), sin32(a32)};
                        v2 m;
                        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m) : "v"(ps), "s"(k));                   // ps sin
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(pc), "s"(k), "v"(m));   // pc cos - ps sin
                    }
                    v2 wa, wb;
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(wa) : "v"(t), "s"(p0));
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0]" : "=v"(wb) : "v"(t), "s"(p0));
                    c[i] = c[i] * wa;
                    c[i2] = c[i2] * wb;
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
    pass1Columns<R, R1, U, LR1>(c);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        constexpr int NB = R1 / 4 - 1;
        float2 a[3], b[NB];
        // W^c and W^{4c} from the table, the other rows as their powers (two loads per column instead of 3 + NB: the table is 8 KB per
        // row and workgroup, and what a workgroup fetches costs as much as what it computes)
        auto sq = [](float2 w) { return float2{w.x * w.x - w.y * w.y, 2.f * w.x * w.y}; };
        auto mul = [](float2 p, float2 q) { return float2{p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x}; };
        if (DIET && u == 1) {
            a[0] = rotateTwiddle(twA[0], kRotA1Cm1, kRotA1Sin);
            if (NB > 0) b[0] = rotateTwiddle(twB[0], kRotA4Cm1, kRotA4Sin);
        } else {
            a[0] = FRONT ? twA[u] : ldg(prm.tw1 + (COLSTEP * u), uint32_t(tid) * 8u * COLLANE);
            if (NB > 0) b[0] = FRONT ? twB[u] : ldg(prm.tw1 + (COLSTEP * u + 3 * RR), uint32_t(tid) * 8u * COLLANE);
        }
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
            if constexpr (PAIRED) {
                // columns 2 tid, 2 tid + 1: waves 0 .. 3 own round 0, waves 4 .. 7 round 1; sixteen ds_write_b128 (wide stores reach the LDS's
                // store rate from one wave per SIMD, ds_write_b64 needs four: MI355X_MICROARCH.md)
                if ((tid >> 8) == r) e1WritePairRun<R1, LR1>(ldsAddress(lds2 + ((2 * tid) & 511)), c);
                return;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int col = COLLANE * tid + COLSTEP * u;
                if ((col >> 9) == r) {
#ifndef SGZ_E1_PAIRED_WRITES   // single ds_write_b64 (hipcc pairs them into ds_write2st64_b64 otherwise: measured 0.5 % slower)
                    const uint32_t wa = ldsAddress(lds2 + (col & 511));
                    if (u == 0) e1WriteRun<R1, LR1, 0>(wa, c);
                    else if (u == 1) e1WriteRun<R1, LR1, (U > 1 ? R1 : 0)>(wa, c);
                    else if (u == 2) e1WriteRun<R1, LR1, (U > 2 ? 2 * R1 : 0)>(wa, c);
                    else e1WriteRun<R1, LR1, (U > 3 ? 3 * R1 : 0)>(wa, c);
#else
#pragma unroll
                    for (int q = 0; q < R1; ++q) lds2[q * 512 + (col & 511)] = c[u * R1 + brev(q, LR1)];
#endif
                }
            }
        };
        const int rd = q1 * 512 + ix;
        v2 lo[R / 2];
        if constexpr (FRONT && !WALK) {
            // the pass-2 twiddle table -> LDS (8.5 KB behind the exchange areas; the map's maxima take the place later)
            if (tid < kTw2Floats / 4) reinterpret_cast<float4 *>(lds + TAB)[tid] = tw2piece;
            if (T < kTw2Floats / 4 && tid < kTw2Floats / 4 - T) reinterpret_cast<float4 *>(lds + TAB)[T + tid] = tw2tail;
        }
        writeRound(0);
        ldsBarrier();
        const uint32_t rdAddr = ldsAddress(lds2 + rd);
        ldsReadRun64<16, R * 8>(lo, rdAddr);                              // c_hi = h
        ldsReadsDone(lo);
        ldsBarrier();
        writeRound(1);
        ldsBarrier();
        {
            v2 hi[16];
            ldsReadRun64<16, R * 8>(hi, rdAddr);                          // c_hi = 16 + h
            ldsReadsDone(hi);
#pragma unroll
            for (int h = 0; h < R / 2; ++h) { c[h + R / 2] = hi[h]; c[h] = lo[h]; }
        }
    }
    ldsBarrier();                                                        // every wave has read exchange 1: the tiles may overwrite it
    RCLK(2);
    // WALK: the recombination's twiddle and the map's table words are requested HERE, where nothing else of this workgroup is on its way
    // through the CU's fetch path: requested where they are used, a late wave's few bytes queue behind the early waves' requests for the
    // next unit's samples and it waits ~10 k clocks for them (tools/phase_clocks.py: pass 3 -> recombined 1.9 k -> 12.4 k clocks)
    // (N = 32768 as well: a workgroup's few bytes otherwise queue behind the sample requests of the workgroup it shares the CU with: cfg2 -1.8 %
    // with the input L2-resident, -0.7 % from HBM)
    constexpr bool EARLYTAB = WALK || (LR1 == 4 && FRONT);
    [[maybe_unused]] float2 wkEarly;
    if constexpr (EARLYTAB) {
        wkEarly = ldg(prm.twPost, uint32_t(q1 + R1 * ix) * 8u);
        mapper.prefetch(tb, tid);
    }
    // -------------------------------------------------------------------------- pass 2 (c_lo = ix): radix R over c_hi
    ditPacked<LR, 0>(c);
#ifndef SGZ_NO_TW_FUSE
    constexpr bool TWFUSE = LR1 >= 4;       // the pass-2 twiddles are applied BEHIND exchange 2, fused into pass 3's first butterfly level (below)
#else
    constexpr bool TWFUSE = false;
#endif
#ifndef SGZ_LATE_DEAD_BARRIER
    constexpr bool EARLYDEAD = TWFUSE && LR1 == 5;     // "the tiles are dead" barrier behind pass 3's table reads instead of behind the recombination (below)
#else
    constexpr bool EARLYDEAD = false;
#endif
    if constexpr (LR1 >= 4 && !TWFUSE) {
        // times W_1024^{c_lo q2}: the whole table sits in LDS behind the exchange areas (copied at the start of the kernel) -- one
        // ds_read_b64 and one complex product per value, instead of 10 fetched rows and 21 products to build the other 21
        const float4 *tab = reinterpret_cast<const float4 *>(lds + TAB + ix * kTw2Row);      // this thread's 32 factors, contiguous: 16 ds_read_b128
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            const float4 w = tab[i];
            if (i > 0) c[brev(2 * i, LR)] = cmul(c[brev(2 * i, LR)], v2{w.x, w.y});
            c[brev(2 * i + 1, LR)] = cmul(c[brev(2 * i + 1, LR)], v2{w.z, w.w});
        }
    } else if constexpr (!TWFUSE) {
        TwFactors<LR> tw;
        tw.load(prm.tw2, ix, R);
        tw.apply(c);
    }
    RCLK(3);
    // -------------------------------------------------------------------------- exchange 2: wave-local R x R transposes
    {
        const int tile = group * TILE;
        const uint32_t rowAddr = ldsAddress(lds + tile + ix * ROW);       // single ds_read_b64s (fft_common.hpp ldsRead64)
#pragma unroll
        for (int q2 = 0; q2 < R; ++q2) lds[tile + q2 * ROW + ix] = c[brev(q2, LR)].x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        { v2 t[16]; ldsReadRun64<16, 8>(t, rowAddr); ldsReadsDone(t);
#pragma unroll
          for (int j = 0; j < R / 2; ++j) { c[2 * j].x = t[j].x; c[2 * j + 1].x = t[j].y; } }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q2 = 0; q2 < R; ++q2) lds[tile + q2 * ROW + ix] = c[brev(q2, LR)].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        { v2 t[16]; ldsReadRun64<16, 8>(t, rowAddr); ldsReadsDone(t);
#pragma unroll
          for (int j = 0; j < R / 2; ++j) { c[2 * j].y = t[j].x; c[2 * j + 1].y = t[j].y; } }
    }
    RCLK(4);
    // -------------------------------------------------------------------------- pass 3 (q2 = ix): radix R over c_lo
    float2 wk;                                                           // W_N^{kc}, for the recombination (a bin's is W_N^{kc} W_{2R}^{m3})
    if constexpr (EARLYTAB) wk = wkEarly; else wk = ldg(prm.twPost, uint32_t(q1 + R1 * ix) * 8u);
    if constexpr (TWFUSE) {
        // The factor W_1024^{c_lo q2} every value still owes from pass 2 is the same number on either side of the transpose (the table is
        // symmetric: this thread is q2 = ix now and register r is c_lo = r), so it is applied HERE, inside pass 3's first level -- whose
        // butterflies (registers r, r + 16) carry no twiddle of their own: u = ta a, p = u + tb b, q = 2u - p is five packed operations
        // where the product in front of the exchange plus a plain butterfly took six (16 of a pass's ~250 fewer).  Same table, same
        // sixteen ds_read_b128 per thread.
        const float4 *tab = reinterpret_cast<const float4 *>(lds + TAB + ix * kTw2Row);
#pragma unroll
        for (int i = 0; i < R / 4; ++i) {
            const float4 wa = tab[i], wb = tab[i + R / 4];               // entries 2i, 2i + 1 and 2i + 16, 2i + 17
            if (i == 0) bflyPackedFusedTw<true>(c[0], c[R / 2], v2{1.f, 0.f}, v2{wb.x, wb.y});
            else bflyPackedFusedTw<false>(c[2 * i], c[2 * i + R / 2], v2{wa.x, wa.y}, v2{wb.x, wb.y});
            bflyPackedFusedTw<false>(c[2 * i + 1], c[2 * i + 1 + R / 2], v2{wa.z, wa.w}, v2{wb.z, wb.w});
        }
        // These were the workgroup's last reads of the exchange tiles and of the table: from here on |X| may overwrite them.  The barrier
        // that says so stood behind the recombination until round 6, where every wave waited for the SLOWEST wave's recombination
        // before its first store (phase clocks of the N = 65536 kernel: 3.6 k of a unit's 50 k clocks between the last wave's
        // recombination and the magnitudes' barrier for 2 k clocks of stores).  Here the waves that arrive early wait while their SIMD
        // is busy with the others' pass 2 anyway, and behind it every wave stores its magnitudes as soon as it has them: N = 65536
        // (one workgroup per CU) -3.6 %; N = 32768, where a second workgroup fills such gaps, +0.3 ... +0.7 % -- so only there
        // (tools/ab.sh, profiles/r06h/ab_early_barrier.txt).
        if constexpr (EARLYDEAD) ldsBarrier();
        ditPacked<LR, 0, R, 2>(c);
    } else {
        ditPacked<LR, 0>(c);
    }
    RCLK(5);
    // Z[kc + T m3] at register brev(m3), kc = q1 + R1 ix
    const int kc = q1 + R1 * ix;
    if (tid == 0) {                                                         // column 0 (k = T m3) pairs registers inside thread 0
#pragma unroll
        for (int m3 = 0; m3 < R; ++m3) {
            lds[SCRATCH + 2 * m3] = c[brev(m3, LR)].x;
            lds[SCRATCH + 2 * m3 + 1] = c[brev(m3, LR)].y;
        }
        // this channel's Nyquist bin X[M] = Re Z[0] - Im Z[0]: csf[N/2] needs both channels' (realLateKernel)
        if (!MONO) prm.ny[self] = 2.f * (c[0].x - c[0].y);                  // (the transform runs on x w / 2)
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
            fixA = lds[SCRATCH] + lds[SCRATCH + 1];                         // csf[0] = Re(csf[0]) * 0.5 / csf[N] = Im(csf[0]) * 0.5 (:861-862): X_c[0] / 2, signed (the 1/2 came in with the window)
            if (MONO) { fixA = __builtin_fabsf(fixA); fixB = lds[SCRATCH] - lds[SCRATCH + 1]; }   // |X[0]| / 2 and X[N/2] / 2 (:547-552)
        }
    }
    // ---- recombination.  a = Z[k] (own register m3 < R/2), b = Z[M - k] (lane L ^ R, register R-1-m3):
    //   2 E = a + conj b,  2 W O = -i w (a - conj b):   2 X[k] = 2E + 2WO   and   2 X[M - k] = conj(2E - 2WO)
    // so ONE evaluation gives the magnitudes of both bins of the pair.  A lane does this for its registers m3 < R/2; their mirrors are
    // the partner lane's registers >= R/2, and the partner does the same for ITS lower registers, whose mirrors are this lane's upper
    // ones: every bin of the lane pair is produced exactly once, with half the permutes, twiddles and adds of bin-by-bin evaluation.
    float magA[R / 2], magB[R / 2];
    [[maybe_unused]] UnitId nextUid = uid;
    [[maybe_unused]] bool more = false;
    [[maybe_unused]] const float *nextX = nullptr;
    [[maybe_unused]] v2 nextRows[R / 2];
    if constexpr (WALK) {
        const uint32_t nextIndex = walkIndex + gridDim.x;
        more = nextIndex < totalUnits;
        if (more) { nextUid = unitOfIndex<MONO>(prm, nextIndex, totalUnits); walkIndex = nextIndex; }
        nextX = samplesOf(nextUid);                                       // (the last unit of a walk re-requests rows of its own: no branch per request)
    }
    {
        // lane holding Z[M - k]: L ^ R, except in slot 0 (q1 = 0: q2' = R - q2 ; q1 = R1/2: q2' = R-1-q2, same half)
        const int lane = int(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
        int plane = lane ^ R;
        if (slot == 0) plane = (lane & ~(R - 1)) | (half ? R - 1 - l : ((R - l) & (R - 1)));
        plane <<= 2;
        auto partnerOf = [&](float v) {
            return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plane, __builtin_bit_cast(int, v)));
        };
        // The upper registers are only read by the partner lane: they are exchanged in place.  Slots >= 1: the partner is lane L ^ 32, two
        // v_permlane32_swap_b32 per value on the vector ALUs (fft_common.hpp halfWaveExchange4), no LDS crossbar.  Slot 0 (q1 = 0 and R1/2)
        // pairs lanes inside each half: ds_bpermute.
#ifdef SGZ_MIRROR_BPERMUTE
        if (true) {
#else
        if (slot == 0) {
#endif
#pragma unroll
            for (int m3 = 0; m3 < R / 2; ++m3) { const int ip = brev(R - 1 - m3, LR); c[ip] = v2{partnerOf(c[ip].x), partnerOf(c[ip].y)}; }
        } else {
#pragma unroll
            for (int m3 = 0; m3 < R / 2; m3 += 4)
                halfWaveExchange4(c[brev(R - 1 - m3, LR)], c[brev(R - 2 - m3, LR)], c[brev(R - 3 - m3, LR)], c[brev(R - 4 - m3, LR)]);
        }
#pragma unroll
        for (int m3 = 0; m3 < R / 2; ++m3) {
            const int i = brev(m3, LR), ip = brev(R - 1 - m3, LR);
            const v2 a = c[i];
            const v2 b = c[ip];
            const v2 w = m3 == 0 ? v2{wk.x, wk.y} : cmulConjK(v2{wk.x, wk.y}, v2{cos64(m3), sin64(m3)});
            // on (re, im) pairs: E = a + conj b, D = a - conj b, O = -i w D = (w.x D.y + w.y D.x, w.y D.y - w.x D.x); then the real parts
            // (E.x + O.x, E.x - O.x) of 2 X[k], 2 X[M - k] in one pair and the imaginary parts (E.y + O.y, E.y - O.y) in another: both
            // squared magnitudes come out of one packed multiply and one packed multiply-add
            v2 E, D, t, O, re2, im2, sq;
            asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(E) : "v"(a), "v"(b));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(D) : "v"(a), "v"(b));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(w), "v"(D));                                   // (w.x D.y, w.y D.y)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(O) : "v"(w), "v"(D), "v"(t));     // (+ w.y D.x, - w.x D.x)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(re2) : "v"(E), "v"(O));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(im2) : "v"(E), "v"(O));
            sq = re2 * re2;
            asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(sq) : "v"(im2), "v"(sq));
            const float pr = re2.x, mr = re2.y, pi = im2.x, mi = im2.y;
            magA[m3] = __builtin_amdgcn_sqrtf(sq.x);                       // (|2 X| / 2: the 1/2 came in with the window)
            magB[m3] = __builtin_amdgcn_sqrtf(sq.y);
            if constexpr (WALK) {
                // registers i and ip are free: one row of the next unit's first sixteen (rows 0 .. 7, 16 .. 23) is requested into them
                const int row = m3 < 8 ? m3 : m3 + 8;
                nextRows[m3] = ldgPinned<v2>(nextX, uint32_t(elemOf(row)) * 8u, lane8);
            }
            if (MONO && m3 == 0 && prm.lowCount[0] && kc >= 1 && kc <= 8) {
                // 2 X[kc] = (pr, pi), 2 X[M - kc] = (mr, -mi):  csf[N - kc] = Z[N - kc] = conj X[kc] (slot 8 - kc),
                // csf[N/2 + kc] = conj X[M - kc] (slot 8 + kc; kc = 8 has none)
                spec[2 * (8 - kc)] = pr; spec[2 * (8 - kc) + 1] = -pi;
                if (kc < 8) { spec[2 * (8 + kc)] = mr; spec[2 * (8 + kc) + 1] = mi; }
            }
        }
    }
    if constexpr (WALK) {
#pragma unroll
        for (int m3 = 0; m3 < R / 2; ++m3) c[m3 < 8 ? m3 : m3 + 8] = nextRows[m3];
    }
    RCLK(6);
    // csf[N/2 - 1] *= 0.5 (quirk Q3, TransformDSP.inl:864): the left channel's bin M - 1 = the mirror of bin 1
    if (!MONO && side == 0 && q1 == 1 && ix == 0) magB[0] *= 0.5f;
    if constexpr (WALK) mapper.arrived(); else if constexpr (!EARLYTAB) mapper.prefetch(tb, tid);
    if constexpr (!EARLYDEAD) ldsBarrier();                              // the tiles are dead: |X| may overwrite them (EARLYDEAD: that barrier stands behind pass 3's table reads, above)
    {
        // left: bin k at position k; right: at position M - k (csf[N - k] = |X_R[k]|: csf order is ascending in LDS on both sides)
        // (two base addresses and compile-time offsets: a run-time stride costs a 64-bit multiply-add per store)
        const int up = chunkPos(kc), down = chunkPos(M - kc);
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
        auto put = [&](int k, float v) { const int i = side ? M - k : k; lds[chunkPos(i)] = v; };
        if (tid >= 1) {
            put(T * tid, fixA);
            if (tid != R / 2) put(T * (R - tid), fixB);
        } else {
            put(0, fixA);
            if (MONO) { spec[16] = fixB; spec[17] = 0.f; }                   // csf[N/2] = Z[N/2] / 2 among the complex entries (slot 8)
            put(M, MONO ? fixB : 0.f);                                      // pairs: csf[N/2] is settled late, 0 can never win meanwhile (strict >)
        }
    }
    // the two pad floats behind this thread's chunk and the floats behind entry M: tap windows read over them with weight 0
    *reinterpret_cast<float2 *>(lds + chunkPos(32 * tid) + 32) = float2{0.f, 0.f};
    if (tid < 16) lds[chunkPos(M) + 1 + tid] = 0.f;
    ldsBarrier();
    RCLK(7);
    realMapSettle<LR1, MIX>(prm, lds, tid, side, task, self, tb, mapper, re, ce, spec);
    if constexpr (!WALK) break;
    else {
        if (!more) break;
        uid = nextUid;
        ldsBarrier();                                                    // the map has read its maxima: the next unit's table and exchanges may overwrite them
    }
    }
}

// Test hook (sgz_stage_map_from_bins on a channel-split plan): csf magnitudes [task][N + 1] come from HBM instead of the transform;
// everything behind them -- pair exchange, chunk-scan map, late-pixel settlement -- is realMapSettle, the code the transform kernel runs.
// csf[N/2] (the left side's last entry, the right side's first) is published as this channel's "Nyquist bin" and taken as it is by
// realLateKernel (prm.binsIn != null switches the |X_L[M] + i X_R[M]| / 2 evaluation off there).  Pairs only.
template <int LR1>
__global__ void __launch_bounds__(1 << (LR1 + 5)) realMapFromBinsKernel(const RealParams prm)
{
    constexpr int R = 32, R1 = 1 << LR1, T = R1 * R, M = R1 * R * R, N = 2 * M;
    constexpr int XFLOATS = realXFloats(M);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const long unit = blockIdx.x;
    const int side = int(unit & 1);
    const long task = unit >> 1;
    const long self = (task << 1) | side;
    const uint32_t maxSlots = prm.chunkSlots[0] > prm.chunkSlots[1] ? prm.chunkSlots[0] : prm.chunkSlots[1];
    const ChunkTables tb{prm.chunkEnds + side * T, prm.chunkReBase + side * T,
                         reinterpret_cast<const uint2 *>(prm.chunkRec) + size_t(side) * prm.P, prm.recs + size_t(side) * prm.P, prm.weights12,
                         prm.mapped ? prm.mapped + (size_t(task) * 2 + side) * prm.P : nullptr, prm.nyBest + size_t(self) * 64, int(prm.fixFrom[side]), int(prm.P), side != 0, nullptr};
    float *re = lds + XFLOATS, *ce = re + maxSlots + 1;
    ChunkMap<T> mapper;
    mapper.prefetch(tb, tid);
    const float *src = prm.binsIn + size_t(task) * (N + 1) + (side ? M : 0);
    for (int i = tid; i <= M; i += T) lds[chunkPos(i)] = (i == (side ? 0 : M)) ? 0.f : src[i];   // pairs: csf[N/2] is settled late, 0 can never win meanwhile
    *reinterpret_cast<float2 *>(lds + chunkPos(32 * tid) + 32) = float2{0.f, 0.f};
    if (tid < 16) lds[chunkPos(M) + 1 + tid] = 0.f;
    if (tid == 0) prm.ny[self] = src[side ? 0 : M];
    __syncthreads();
    realMapSettle<LR1, 0>(prm, lds, tid, side, task, self, tb, mapper, re, ce, lds);
}

// The pixels of a frame that need both channels, behind the launch whose workgroups each saw one:
//   * csf[N/2] = |X_L[M] + i X_R[M]| / 2 (TransformDSP.inl:863) is the LAST offset of the arg-max scans of either side's top pixels
//     (fixFrom .. P), compared with a strict >: it wins exactly when its square exceeds the winning square the channel's workgroup
//     left in nyBest; the pixel then shows csf[N/2] itself;
//   * the interpolated pixels whose tap window reaches over bin 0 (..., csf[N-1], csf[N], csf[0], csf[1], ...: the other channel's
//     lowest bins): evaluated here from the kLowBins lowest entries both channels left in `low`, taps in order.
// One thread per pixel: items [0, tasks * 2 * nTop) are the top pixels (task, side, j), the rest the low pixels (task, n) -- a few
// hundred threads at cfg2, so that the launch costs little more than its boundary.
__host__ __device__ LateFix lateFixOf(const RealParams &prm)
{
    return LateFix{prm.ny, prm.nyBest, prm.fixFrom[0], prm.fixFrom[1], prm.P, prm.invSize, prm.binsIn ? 1u : 0u};
}
__global__ void __launch_bounds__(256) realLateKernel(const RealParams prm, const int N, const uint32_t nTop)
{
#pragma clang fp contract(off)
    const long tasks = prm.frames * long(prm.C);
    const long item = long(blockIdx.x) * 256 + threadIdx.x;
    const long tops = tasks * 2 * nTop;
    const uint32_t nLow = prm.lowCount[0] + prm.lowCount[1];
    if (item < tops) {
        const long task = item / (2 * nTop);
        const int s = int((item / nTop) & 1);
        const uint32_t j = uint32_t(item % nTop);
        const LateFix lf = lateFixOf(prm);
        const uint32_t x = lf.fixFrom(s) + j;
        const bool top = x < prm.P && prm.mapped;
        // every load goes out before anything is looked at: one memory round trip
        float *px = prm.mapped + size_t(2 * task + s) * prm.P + (top ? x : 0u);
        const float own = top ? *px : 0.f;
        const float best = top ? lateBestSquare(lf, task, s, x) : 0.f;
        const float vM = lateNyquistBin(lf, task);
        if (s == 0 && j == 0 && prm.binsOut) prm.binsOut[size_t(task) * (N + 1) + N / 2] = vM;
        if (top) {
            const float v = lateNyquistPixel(lf, vM, best, own);
            if (v != own || !(own == own)) *px = v;
        }
    } else if (item < tops + tasks * nLow && prm.mapped) {
        const long task = (item - tops) / nLow;
        const uint32_t n = uint32_t((item - tops) % nLow);
        const int s = n < prm.lowCount[0] ? 0 : 1;
        const uint32_t x = prm.lowPixels[n];
        const PixelRec rec = prm.recsFull[size_t(s) * prm.P + x];
        const float *lowL = prm.low + size_t(2 * task) * kLowBins, *lowR = lowL + kLowBins;
        float acc = 0.f;
        int k = rec.a;
        for (int i = 0; i < rec.b; ++i) {                                    // taps in order (lanczosFilter restatement)
            // csf[k]: k < kLowBins is the left channel's bin k, k > N - kLowBins the right channel's bin N - k
            const float v = k < kLowBins ? lowL[k] : lowR[N - k];
            acc = acc + v * prm.weights[rec.c + i];
            k = (k == N) ? 0 : k + 1;
        }
        prm.mapped[(size_t(2 * task + s)) * prm.P + x] = finishPixel<5>(prm.invSize * acc);
    }
}
hipError_t launchRealLate(const RealParams &prm, uint32_t N, hipStream_t stream)
{
    const long tasks = prm.frames * long(prm.C);
    if (tasks <= 0) return hipSuccess;
    uint32_t nTop = 1;                                                       // (>= 1: item (task, 0, 0) also writes binsOut[N/2])
    for (int s = 0; s < 2; ++s) nTop = std::max(nTop, std::min(64u, prm.P - std::min(prm.P, prm.fixFrom[s])));
    const long items = tasks * 2 * nTop + tasks * long(prm.lowCount[0] + prm.lowCount[1]);
    hipLaunchKernelGGL(realLateKernel, dim3(unsigned((items + 255) / 256)), dim3(256), 0, stream, prm, int(N), nTop);
    return hipGetLastError();
}

// test hook: y = finishPixel(x), the last step of every K_A pixel (stft_body.hpp)
__global__ void __launch_bounds__(256) finishPixelKernel(const float *x, float *y, size_t n)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) y[i] = finishPixel<5>(x[i]);
}
hipError_t launchFinishPixel(const float *x, float *y, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(finishPixelKernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, x, y, n);
    return hipGetLastError();
}

hipError_t launchStftReal(const RealParams &prm, uint32_t N, hipStream_t stream)
{
    const bool mono = prm.mode != SGZ_CH_SEPARATE && prm.mode != SGZ_CH_MIDSIDE;
    const long units = prm.frames * long(prm.C) * (mono ? 1 : 2);
    if (units <= 0) return hipSuccess;
    // N = 32768, pairs: the 1024-thread form (spectrum_real16.hip) unless the plan switched it off or its LDS layout does not fit
    if (N == 32768 && !mono && prm.tw16 && real16LdsBytes(std::max(prm.chunkSlots[0], prm.chunkSlots[1])) <= 80 * 1024) return launchStftReal16(prm, stream);
    const uint32_t M = N / 2, T = M / 32;
    const uint32_t maxSlots = std::max(prm.chunkSlots[0], prm.chunkSlots[1]);
    // magnitudes, then the map's tile / chunk maxima (the same floats hold column 0's scratch during the recombination), then (mono) the complex entries
    const size_t ldsBytes = (size_t(realXFloats(int(M))) + realExtraFloats(maxSlots, T, N >= 32768) + 64 + (mono ? 2 * kSpecBins : 0)) * 4;
    static size_t granted[27][64] = {};
    const bool wcos = prm.winPhase != nullptr;
    if (prm.binsIn) {
        if (mono) return hipErrorNotSupported;
        auto inject = [&](auto kern, int slot, unsigned threads) -> hipError_t {
            if (hipError_t e = grantLds(reinterpret_cast<const void *>(kern), ldsBytes, granted[slot]); e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(unsigned(units)), dim3(threads), ldsBytes, stream, prm);
            return launchRealLate(prm, N, stream);
        };
        if (N == 32768) return inject(&realMapFromBinsKernel<4>, 24, 512);
        if (N == 16384) return inject(&realMapFromBinsKernel<3>, 25, 256);
        if (N == 65536) return inject(&realMapFromBinsKernel<5>, 26, 1024);
        return hipErrorNotSupported;
    }
    auto go = [&](auto kern, int slot, unsigned threads, size_t limit) -> hipError_t {
        if (ldsBytes > limit) return hipErrorInvalidValue;
        if (hipError_t e = grantLds(reinterpret_cast<const void *>(kern), ldsBytes, granted[slot]); e != hipSuccess) return e;
        // (N = 65536, pairs, window evaluated in the kernel: one workgroup per CU walks over the units -- WALK in the kernel)
        const bool walk = N == 65536 && !mono && prm.mode == SGZ_CH_SEPARATE && wcos && prm.roundSize;
        hipLaunchKernelGGL(kern, dim3(unsigned(walk ? std::min<long>(units, long(prm.roundSize)) : units)), dim3(threads), ldsBytes, stream, prm);
        // the pixels that need both channels (skipped when the caller's next kernel overlays them itself: prm.lateInNext)
        if (!mono && !prm.lateInNext) return launchRealLate(prm, N, stream);
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
