// resonator.hip -- the Spectrum view's second transform algorithm, SpectrumContent::TransformAlgorithm::RSNT ("Resonator").
//
// Reference: the audio thread feeds every sample to a bank of complex one-pole resonators, one per axis point and "vector"
// (TransformPair::resonatingDispatch, Source/Spectrum/TransformDSP.inl:1213-1295 -> cpl::dsp::CComplexResonator::resonateReal); every
// sampleBufferSize samples a frame is the windowed state (audioEntryPoint :1172-1201 -> mapToLinearSpace's RSNT branch :1103-1133 ->
// CComplexResonator::getWholeWindowedState), which then runs through the same mapAndTransformDFTFilters / blend stages as an FFT
// frame (K_B, spectrum_post.hip).  CComplexResonator lives in the absent cpl submodule: its arithmetic is restated from the
// mathematics it implements (plan.cpp buildResonator; oracle/resonator.c is the checker and states the same choices).
//
// MI355X form.  The recurrence  s[n] = c s[n-1] + x[n]  is sequential in time per resonator but linear, so time is cut at the frame
// boundaries.  A ONE-frame launch (the real-time case) continues the carried state sample by sample on the vector ALUs,
// contraction-free: the reference's recurrence step for step.  In a launch of several frames every frame starts from rest and the
// frames are chained afterwards:  s_f = c^hop s_{f-1} + local_f  (c^hop in double from the fp32 pole, carried as hi + lo words).
// The frames from rest are block sums against the pole's powers -- a matrix product.  When hop is a multiple of 1024 they run on the
// matrix cores: resonateMfmaBf16Kernel (default: every fp32 sample and weight as the exact sum of three bfloat16 parts, six part
// products on v_mfma_f32_32x32x16_bf16, fp32-equivalent accuracy at 2.7x the fp32 matrix rate) or resonateMfmaKernel (fp32 matrix
// cores), then a dot product with the powers of pole^32 per 1024-sample tile; the chain is cut into segments of frames that run in
// parallel (resonatorSegmentKernel / SegmentFoldKernel / ChainWindowKernel, the window and the magnitudes behind every chain step).
// Any other hop: the vector ALUs eight samples per step (resonateKernel's block form; frame 0 of the launch continues the carried
// state exactly), resonatorChainKernel, resonatorWindowKernel.  Against the sequential fp32 recurrence the chained result differs by
// the roundings of a random walk over the resonator's memory (tests/test_gpu_resonator.py states the bar).  Bytes: 8 hop per
// (frame, pair) in, 8 V P per (frame, signal) through HBM between the kernels: the path is compute-bound.
#include "kernels.hpp"

#include <hip/hip_runtime.h>

namespace sgz {

namespace {

constexpr int kResBlock = 256;

// TransformPair::resonatingDispatch (TransformDSP.inl:1250-1293): which signal a resonator bank sees
__device__ __forceinline__ float resMix(uint32_t mode, int signal, float l, float r)
{
    switch (mode) {
    case SGZ_CH_RIGHT: return r;
    case SGZ_CH_LEFT: return l;
    case SGZ_CH_MERGE: return l + r;
    case SGZ_CH_SIDE: return l - r;
    case SGZ_CH_MIDSIDE: return signal == 0 ? l - r : l + r;
    default: return signal == 0 ? l : r;
    }
}

// one sample into the V chains of a thread.  EXACT: the reference's recurrence operation by operation (no contraction) -- the frame that
// continues the carried state; otherwise fused multiply-adds (4 operations instead of 7): the frames that start from rest are chained
// with c^hop afterwards and are held to a tolerance, not to the sequential recurrence's bits
template <int V, bool EXACT>
__device__ __forceinline__ void resStep(float (&re)[V], float (&im)[V], const float (&cr)[V], const float (&ci)[V], float x)
{
    if constexpr (EXACT) {
#pragma clang fp contract(off)
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float nre = (re[v] * cr[v] - im[v] * ci[v]) + x;
            const float nim = re[v] * ci[v] + im[v] * cr[v];
            re[v] = nre; im[v] = nim;
        }
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float nre = __builtin_fmaf(re[v], cr[v], __builtin_fmaf(-im[v], ci[v], x));
            const float nim = __builtin_fmaf(re[v], ci[v], im[v] * cr[v]);
            re[v] = nre; im[v] = nim;
        }
    }
}

// samples per block step of the frames that start from rest: B consecutive steps of the recurrence collapse into
//     s' = c^B s + sum_b c^(B-1-b) x[b]                (4 + 2 (B - 1) + 1 fused operations per B samples instead of 4 B),
// the powers c^1 .. c^B living in registers (2 B per vector: B = 8 up to five vectors, 4 beyond)
template <int V> struct ResBlock { static constexpr int B = V <= 5 ? 8 : 4; };

template <int V, bool EXACT>
__device__ __forceinline__ void resRun(const ResParams &prm, float *xs, const float *L, const float *R, int signal, int tid, uint32_t i, bool live,
                                       float (&re)[V], float (&im)[V], const float (&cr)[V], const float (&ci)[V])
{
    constexpr int B = ResBlock<V>::B;
    [[maybe_unused]] float pr[V][B], pi[V][B];                    // pr[v][k] + i pi[v][k] = c_v^(k+1), from the plan (rounded once each)
    [[maybe_unused]] float lr[V], li[V];                          // low words of c_v^B: the state is multiplied by hi + lo
    if constexpr (!EXACT) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float2 lo = prm.cpowBLo[(size_t(v) * prm.P + (live ? i : 0u)) * 2 + (B == 8 ? 1 : 0)];
            lr[v] = lo.x; li[v] = lo.y;
            const float4 *q = reinterpret_cast<const float4 *>(prm.cpowB + (size_t(v) * prm.P + (live ? i : 0u)) * 8);
#pragma unroll
            for (int k = 0; k < B; k += 2) {
                const float4 t = q[k / 2];
                pr[v][k] = t.x; pi[v][k] = t.y; pr[v][k + 1] = t.z; pi[v][k + 1] = t.w;
            }
        }
    }
    for (uint32_t t0 = 0; t0 < prm.hop; t0 += kResBlock) {
        const uint32_t n = min(uint32_t(kResBlock), prm.hop - t0);
        __syncthreads();
        xs[tid] = uint32_t(tid) < n ? resMix(prm.mode, signal, L[t0 + tid], R[t0 + tid]) : 0.f;
        __syncthreads();
        if (n == uint32_t(kResBlock)) {
            if constexpr (EXACT) {
#pragma unroll 2
                for (int j = 0; j < kResBlock; j += 4) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(xs + j);
                    resStep<V, true>(re, im, cr, ci, x4.x);
                    resStep<V, true>(re, im, cr, ci, x4.y);
                    resStep<V, true>(re, im, cr, ci, x4.z);
                    resStep<V, true>(re, im, cr, ci, x4.w);
                }
            } else {
                for (int j = 0; j < kResBlock; j += B) {
                    float x[B];
#pragma unroll
                    for (int k = 0; k < B; k += 4) {
                        const float4 x4 = *reinterpret_cast<const float4 *>(xs + j + k);
                        x[k] = x4.x; x[k + 1] = x4.y; x[k + 2] = x4.z; x[k + 3] = x4.w;
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        // c^B s, c^B = hi + lo
                        float nre = __builtin_fmaf(re[v], pr[v][B - 1], __builtin_fmaf(-im[v], pi[v][B - 1], __builtin_fmaf(re[v], lr[v], -im[v] * li[v])));
                        float nim = __builtin_fmaf(re[v], pi[v][B - 1], __builtin_fmaf(im[v], pr[v][B - 1], __builtin_fmaf(re[v], li[v], im[v] * lr[v])));
                        // + sum_b c^(B-1-b) x[b]   (x is real; c^0 = 1)
#pragma unroll
                        for (int b = 0; b < B - 1; ++b) {
                            nre = __builtin_fmaf(x[b], pr[v][B - 2 - b], nre);
                            nim = __builtin_fmaf(x[b], pi[v][B - 2 - b], nim);
                        }
                        re[v] = nre + x[B - 1];
                        im[v] = nim;
                    }
                }
            }
        } else {
            for (uint32_t j = 0; j < n; ++j) resStep<V, EXACT>(re, im, cr, ci, xs[j]);
        }
    }
}

template <int V>
__global__ __launch_bounds__(kResBlock) void resonateKernel(ResParams prm)
{
    __shared__ __attribute__((aligned(16))) float xs[kResBlock];
    const int tid = threadIdx.x;
    const uint32_t i = blockIdx.x * kResBlock + tid;
    const bool live = i < prm.P;
    const uint32_t unit = blockIdx.y;                              // (frame, pair, signal)
    const int signal = int(unit % uint32_t(prm.signals));
    const uint32_t pair = (unit / uint32_t(prm.signals)) % prm.C;
    const long frame = long(unit / (uint32_t(prm.signals) * prm.C));
    const float *L = prm.planar + size_t(2 * pair) * prm.chStride + size_t(frame) * prm.hop;
    const float *R = L + prm.chStride;
    const bool continues = frame == 0 && prm.firstContinues;       // uniform per workgroup

    float cr[V], ci[V], re[V], im[V];
    const size_t stateAt = (size_t(pair) * 2 + size_t(signal)) * V * prm.P + i;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const float2 c = live ? prm.coeff[size_t(v) * prm.P + i] : float2{0.f, 0.f};
        cr[v] = c.x; ci[v] = c.y;
        const float2 s0 = (live && continues) ? prm.state[stateAt + size_t(v) * prm.P] : float2{0.f, 0.f};
        re[v] = s0.x; im[v] = s0.y;
    }
    if (continues) resRun<V, true>(prm, xs, L, R, signal, tid, i, live, re, im, cr, ci);
    else resRun<V, false>(prm, xs, L, R, signal, tid, i, live, re, im, cr, ci);
    if (live) {
        float2 *out = prm.local + ((size_t(frame) * prm.C + pair) * size_t(prm.signals) + size_t(signal)) * V * prm.P + i;
#pragma unroll
        for (int v = 0; v < V; ++v) out[size_t(v) * prm.P] = float2{re[v], im[v]};
    }
}

// ---- the frames from rest on the matrix cores.  A frame's `hop` samples are tiles of 32 blocks x 32 samples; for one tile and 32
// resonators n the block sums are a matrix product
//     Y[a][n] = sum_b x[32 a + b] pole_n^(31 - b),          a, b < 32
// i.e. D (32 x 32) += A (32 x 2: two samples of every block) x B (2 x 32: the 32 resonators' weights of those two samples), sixteen
// v_mfma_f32_32x32x2_f32 per real / imaginary part (exact fp32 fused multiply-add chains).  The tile's contribution to the state is
//     T[n] = sum_a Y[a][n] pole_n^(32 (31 - a)),            state <- pole^1024 state + T
// (a lane holds 16 of a resonator's 32 block sums -- rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of D's column lane & 31 -- and the lane
// 32 further the others: each takes its partial dot product, one cross-lane add joins them).  The weights are plan tables (double,
// rounded once); pole^1024 is carried as hi + lo words like pole^hop.  One wave = 32 resonators of one vector; a 256-thread workgroup =
// 4 such groups sharing the tile's samples in LDS.  Per tile and wave: 32 MFMAs (2 048 issue clocks) against ~100 vector instructions.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef SGZ_RES_WAVES
#define SGZ_RES_WAVES 4
#endif
#ifndef SGZ_RES_UPW
#define SGZ_RES_UPW 4
#endif
constexpr unsigned kResUnitsPerWg = SGZ_RES_UPW;                     // units (frame, pair, signal) a workgroup of the bf16 form walks in a row
// Behind every matrix instruction of the bf16 kernel: `s_nop SGZ_RES_EXP_GAP` (default 3 = 4 idle cycles) between two sched_barriers.
// Round 6 (NOTES.md "A matrix-core kernel that disturbs its neighbours"): while this kernel runs, FFT kernels elsewhere on the device come
// back with wrong cache lines -- rarely without the s_nop (1-4 of 24 000 launches), in 60 % of the launches with s_nop 9 ... 15 or in the
// compiler's own order (-DSGZ_RES_NO_SCHED), and in NONE of 8 x 24 000 with s_nop 1 ... 7 (+3 ... +7 % of the render; profiles/r06g/
// rsnt_gap_scan.txt).  The middle of the clean range is shipped; the kernel stays opt-in all the same (sgz.h SGZ_OPT_MATRIX_RESONATOR).
#ifndef SGZ_RES_EXP_GAP
#define SGZ_RES_EXP_GAP 3
#endif
#if defined(SGZ_RES_NO_SCHED)                                        // (platform experiment: let the compiler order the tile loop)
#define SGZ_RES_SCHED_BARRIER() do { } while (0)
#elif SGZ_RES_EXP_GAP >= 0
#define SGZ_RES_SCHED_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" ::"n"(SGZ_RES_EXP_GAP)); __builtin_amdgcn_sched_barrier(0); } while (0)
#else                                                               // (-DSGZ_RES_EXP_GAP=-1: rounds 4-5's stream, no idle cycles)
#define SGZ_RES_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef SGZ_RES_BF16_OCC
#define SGZ_RES_BF16_OCC 3                                          // waves per SIMD the bf16 form is compiled for (168 registers)
#endif

// WAVES: (vector, group) waves per workgroup -- they share the tile's samples in LDS and meet at one barrier per tile
template <int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, 4) void resonateMfmaKernel(ResParams prm, int V)
{
    constexpr int TH = 64 * WAVES, PER = 1024 / TH;                  // threads, samples a thread stages per tile
    __shared__ float xs[2 * 32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const uint32_t groups = (prm.P + 31) / 32;                       // resonator groups per vector
    const uint32_t g = blockIdx.x * WAVES + wave;                    // (vector, group) of this wave
    const bool liveWave = g < uint32_t(V) * groups;
    const uint32_t v = liveWave ? g / groups : 0u;
    const uint32_t i = liveWave ? (g - v * groups) * 32 + (lane & 31) : 0u;
    const bool live = liveWave && i < prm.P;
    const uint32_t unit = blockIdx.y;                                // (frame, pair, signal)
    const int signal = int(unit % uint32_t(prm.signals));
    const uint32_t pair = (unit / uint32_t(prm.signals)) % prm.C;
    const long frame = long(unit / (uint32_t(prm.signals) * prm.C));
    const float *L = prm.planar + size_t(2 * pair) * prm.chStride + size_t(frame) * prm.hop;
    const float *R = L + prm.chStride;
    const size_t at = (size_t(v) * prm.P + (live ? i : 0u));
    // this lane's weights: samples 2 s + h of a block (B operand of step s), blocks (r & 3) + 8 (r >> 2) + 4 h of a tile
    float w1r[16], w1i[16], w2r[16], w2i[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const size_t VP = size_t(V) * prm.P;                             // tables are [32][V][P]: coalesced across the lanes
        const float2 q = live ? prm.w1[size_t(2 * s + h) * VP + at] : float2{0.f, 0.f};
        w1r[s] = q.x; w1i[s] = q.y;
        const int a = (s & 3) + 8 * (s >> 2) + 4 * h;
        const float2 u = live ? prm.w2[size_t(a) * VP + at] : float2{0.f, 0.f};
        w2r[s] = u.x; w2i[s] = u.y;
    }
    const float4 tp = live ? prm.tilePow[at] : float4{0.f, 0.f, 0.f, 0.f};
    float sre = 0.f, sim = 0.f;
    // the tile's samples: fetched one tile ahead into registers, parked in one of two LDS buffers (one barrier per tile)
    // (raw left / right values: the channel mix is applied when they are parked -- mixing at the load makes the wave wait for the
    // loads right there, a memory round trip per tile in front of the matrix products: the kernel ran at half the pipe's rate)
    float nl[PER], nr[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { const uint32_t e = uint32_t(tid) + uint32_t(TH) * k; nl[k] = L[e]; nr[k] = R[e]; }
    int buf = 0;
    for (uint32_t t0 = 0; t0 < prm.hop; t0 += 1024, buf ^= 1) {
        float *xb = xs + buf * (32 * 33);
#pragma unroll
        for (int k = 0; k < PER; ++k) { const uint32_t e = uint32_t(tid) + uint32_t(TH) * k; xb[(e >> 5) * 33 + (e & 31)] = resMix(prm.mode, signal, nl[k], nr[k]); }
        __syncthreads();                                             // (the other buffer was read two tiles ago: every wave is past it)
        if (t0 + 1024 < prm.hop) {
#pragma unroll
            for (int k = 0; k < PER; ++k) { const uint32_t e = t0 + 1024 + uint32_t(tid) + uint32_t(TH) * k; nl[k] = L[e]; nr[k] = R[e]; }
        }
        f32x16 dre = {0}, dim = {0};
        const float *arow = xb + (lane & 31) * 33 + h;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = arow[2 * s];
            dre = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w1r[s], dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w1i[s], dim, 0, 0, 0);
        }
        float pr = 0.f, pi = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pr = __builtin_fmaf(dre[r], w2r[r], __builtin_fmaf(-dim[r], w2i[r], pr));
            pi = __builtin_fmaf(dre[r], w2i[r], __builtin_fmaf(dim[r], w2r[r], pi));
        }
        pr += __shfl_xor(pr, 32);
        pi += __shfl_xor(pi, 32);
        const float nre = __builtin_fmaf(sre, tp.x, __builtin_fmaf(-sim, tp.y, __builtin_fmaf(sre, tp.z, -sim * tp.w))) + pr;
        const float nim = __builtin_fmaf(sre, tp.y, __builtin_fmaf(sim, tp.x, __builtin_fmaf(sre, tp.w, sim * tp.z))) + pi;
        sre = nre; sim = nim;
    }
    if (live && h == 0)
        prm.local[((size_t(frame) * prm.C + pair) * size_t(prm.signals) + size_t(signal)) * V * prm.P + size_t(v) * prm.P + i] = float2{sre, sim};
}

// ---- the same block sums on the bfloat16 matrix cores, sixteen times the fp32 form's rate, without giving up fp32 accuracy: every
// fp32 value is the EXACT sum of three bfloat16 (x = h + m + l: 24 significant bits = 3 x 8, each part the round-to-nearest of what
// the parts before left; the last residual has at most eight significant bits), so
//     x w = hh + (hm + mh) + (hl + lh + mm) + [ml + lm + ll  <  2^-23 |x w|: dropped],
// six v_mfma_f32_32x32x16_bf16 products with exact bf16 x bf16 terms accumulated in fp32 -- 24 MFMAs of 32 cycles per tile instead of
// 32 of 64 cycles, small terms first.  A operand: lane l holds block a = l & 31, samples k = 16 kh + 8 (l >> 5) + e (e < 8) of the
// block; B operand: resonator n = l & 31, the weights of the same samples (the contraction index only has to agree between the two).
// The samples are split when they are parked in LDS (three bf16 planes, rows padded to 80 bytes: a wave's ds_read_b128 touch every
// bank once), the weights once per workgroup from the same fp32 table as the fp32 form's.  Everything behind the block sums (the dot
// product with the powers of pole^32, the tile recurrence) is the fp32 form's.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// resMix as  x = cl l + cr r  with cl, cr in {0, 1, -1}: the products are exact and the sum rounds once, as the reference's l + r / l - r
__device__ __forceinline__ void resMixCoeff(uint32_t mode, int signal, float &cl, float &cr)
{
    switch (mode) {
    case SGZ_CH_RIGHT: cl = 0.f; cr = 1.f; break;
    case SGZ_CH_LEFT: cl = 1.f; cr = 0.f; break;
    case SGZ_CH_MERGE: cl = 1.f; cr = 1.f; break;
    case SGZ_CH_SIDE: cl = 1.f; cr = -1.f; break;
    case SGZ_CH_MIDSIDE: cl = 1.f; cr = signal == 0 ? -1.f : 1.f; break;
    default: cl = signal == 0 ? 1.f : 0.f; cr = signal == 0 ? 0.f : 1.f; break;
    }
}

// The loop is software-pipelined by hand.  Back-to-back MFMAs of one wave wait in the SIMD's issue stage for the matrix pipe and
// nothing else issues meanwhile (first form of this kernel: MFMA pipe 51 % busy, vector issue 58 %, the two adding up instead of
// overlapping), so every MFMA is followed by a handful of independent vector instructions of the SAME wave: while the twelve
// products of the real block sums run, the previous tile's imaginary block sums are folded into the state; while the twelve of the
// imaginary sums run, the next tile's samples are split and parked and this tile's real sums are folded (sched_barrier pins the
// order).  The two 32-lane halves of a wave hold 16 block sums each of every resonator: the tile recurrence is linear, so each half
// carries its own partial state and the halves meet once per unit.  A workgroup walks `unitsPerWg` consecutive units (frame, pair,
// signal) as ONE stream of tiles -- the weights are fetched once and the pipeline never drains between units (samples of the next
// unit are requested and parked while the last tiles of this one are multiplied).
// SINGLE: every signal is one input channel as it is (every mode but Merge, Side, MidSide): only that channel is requested, nothing is mixed
template <int WAVES = 4, bool SINGLE = false>
__global__ __launch_bounds__(64 * WAVES, SGZ_RES_BF16_OCC) void resonateMfmaBf16Kernel(ResParams prm, int V, uint32_t nUnits, uint32_t unitsPerWg)
{
    constexpr int TH = 64 * WAVES, PER = 1024 / TH;                  // threads, samples a thread stages per tile
    constexpr int ROW = 40;                                          // uint16 per padded row of 32 samples (80 bytes)
#ifdef SGZ_RES_EXP_DWORD_STORES                                      // (platform experiment, results meaningless: the parts stored as whole dwords)
    using XsWord = uint32_t;
#else
    using XsWord = uint16_t;
#endif
    __shared__ __attribute__((aligned(16))) XsWord xs[2][3][32 * ROW];
#ifdef SGZ_RES_EXP_JITTER                                            // (platform experiment: the workgroups' tile loops out of phase with one another)
    for (uint32_t d = (blockIdx.x * 5u + blockIdx.y * 11u) & 15u; d > 0; --d) asm volatile("s_nop 7");
#endif
#ifdef SGZ_RES_PAD_VGPR                                              // (platform experiments: a register count that allows two waves per SIMD only)
    asm volatile("v_mov_b32 v191, 0" ::: "v191");
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const uint32_t groups = (prm.P + 31) / 32;
    const uint32_t g = blockIdx.x * WAVES + wave;
    const bool liveWave = g < uint32_t(V) * groups;
    const uint32_t v = liveWave ? g / groups : 0u;
    const uint32_t i = liveWave ? (g - v * groups) * 32 + (lane & 31) : 0u;
    const bool live = liveWave && i < prm.P;
    const size_t at = (size_t(v) * prm.P + (live ? i : 0u));
    const size_t VP = size_t(V) * prm.P;
    // this lane's first-level weights as B operands, [part][kh], real and imaginary: split into bf16 parts by the plan (plan.cpp)
    uint4 wr[3][2], wi[3][2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const size_t row = size_t(((kh * 2 + h) * 3 + p) * 2);
            wr[p][kh] = live ? prm.w1b[row * VP + at] : uint4{0u, 0u, 0u, 0u};
            wi[p][kh] = live ? prm.w1b[(row + 1) * VP + at] : uint4{0u, 0u, 0u, 0u};
        }
    float w2r[16], w2i[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int a = (s & 3) + 8 * (s >> 2) + 4 * h;
        const float2 u = live ? prm.w2[size_t(a) * VP + at] : float2{0.f, 0.f};
        w2r[s] = u.x; w2i[s] = u.y;
    }
    const float4 tp = live ? prm.tilePow[at] : float4{0.f, 0.f, 0.f, 0.f};
    const uint32_t tiles = prm.hop / 1024u;
    const uint32_t u0 = blockIdx.y * unitsPerWg, u1 = min(nUnits, u0 + unitsPerWg);
    const uint32_t nT = (u1 - u0) * tiles;                           // the workgroup's stream of tiles
    // where a unit's samples are and how its signal mixes them (uniform: scalar registers)
    auto enter = [&](uint32_t unit, const float *&pl, const float *&pr2, float &cl, float &cr) {
        const int signal = int(unit % uint32_t(prm.signals));
        const uint32_t pair = (unit / uint32_t(prm.signals)) % prm.C;
        const long frame = long(unit / (uint32_t(prm.signals) * prm.C));            // (of this launch: planar / local point at its first frame)
        pl = prm.planar + size_t(2 * pair) * prm.chStride + size_t(frame) * prm.hop;        // (uniform: a scalar base, the lane's offset is tid)
        pr2 = pl + prm.chStride;
        resMixCoeff(prm.mode, signal, cl, cr);
        if constexpr (SINGLE) { if (cl == 0.f) pl = pr2; }           // (the one channel this signal is)
    };
    // a sample's three parts by truncation (the high halves of x, x - h, x - h - m: exact as well, and the stores take the high half
    // of a register as it is); the weights' parts are rounded, so the dropped cross terms carry no sign of their own
    const uint32_t slot = uint32_t(tid >> 5) * ROW + uint32_t(tid & 31);            // sample tid + TH k of a tile: row (tid >> 5) + 8 k
    // the request stream runs two tiles ahead of the products, the parking one tile ahead; the registers nl / nr hold the tile
    // between the two, (clHeld, crHeld) the mix of the unit it belongs to
    const float *pl, *pr2;
    float clHeld, crHeld, clNext, crNext;
    uint32_t uReq = u0, tReq = 0;
    enter(uReq, pl, pr2, clHeld, crHeld);
    float nl[PER], nr[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { nl[k] = pl[tid + TH * k]; if constexpr (!SINGLE) nr[k] = pr2[tid + TH * k]; }
    auto advance = [&]() {                                           // the request stream one tile on (stays on the last tile at the end)
        clNext = clHeld; crNext = crHeld;
        if (tReq + 1 < tiles) { ++tReq; pl += 1024; pr2 += 1024; }
        else if (uReq + 1 < u1) { ++uReq; tReq = 0; enter(uReq, pl, pr2, clNext, crNext); }
    };
#pragma unroll
    for (int k = 0; k < PER; ++k) {                                  // tile 0 parked
        const float x = SINGLE ? nl[k] : __builtin_fmaf(crHeld, nr[k], clHeld * nl[k]);
        const float r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
        const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        const uint32_t o = slot + uint32_t(k) * (TH / 32) * ROW;
        xs[0][0][o] = XsWord(__float_as_uint(x) >> 16); xs[0][1][o] = XsWord(__float_as_uint(r1) >> 16); xs[0][2][o] = XsWord(__float_as_uint(r2) >> 16);
    }
    advance();
    clHeld = clNext; crHeld = crNext;
#pragma unroll
    for (int k = 0; k < PER; ++k) { nl[k] = pl[tid + TH * k]; if constexpr (!SINGLE) nr[k] = pr2[tid + TH * k]; }     // tile 1 held
    float sre = 0.f, sim = 0.f, pr = 0.f, pi = 0.f;                  // this half's partial state; the open tile's partial sums
    f32x16 dre = {0}, dim = {0};                                     // ("tile -1": zero block sums, folded like any other)
    const bf16x8 *Wr = reinterpret_cast<const bf16x8 *>(&wr[0][0]), *Wi = reinterpret_cast<const bf16x8 *>(&wi[0][0]);
    // (sample part, weight part): the terms of 2^-16 first, then 2^-8, then the leading one
    constexpr int TP[6] = {2, 0, 1, 1, 0, 0}, TQ[6] = {0, 2, 1, 0, 1, 0};
    auto finish = [&](uint32_t unit) {                               // a unit's state: the halves together, out, and from rest again
        const float fre = sre + __shfl_xor(sre, 32), fim = sim + __shfl_xor(sim, 32);
        if (live && h == 0) prm.local[size_t(unit) * V * prm.P + size_t(v) * prm.P + i] = float2{fre, fim};
        sre = 0.f; sim = 0.f;
    };
    uint32_t uMul = u0, tMul = 0;                                    // the unit and tile the products are at
    for (uint32_t T = 0; T < nT; ++T) {
        const int buf = int(T & 1u);
        const bool unitEnded = tMul == 0 && T > 0;                   // the tile folded in phase A closes the unit before this one
        advance();                                                   // (scalar: where tile T + 2 is, and how it mixes)
        __syncthreads();                                             // tile T is parked (its other buffer was read a tile ago: every wave is past it)
        bf16x8 A[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
                A[p][kh] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(&xs[buf][p][(lane & 31) * ROW + 16 * kh + 8 * h]));
        // ---- phase A: the real block sums of tile T  |  the imaginary block sums of tile T - 1 into the partial sums, the state onwards
        // (sched_barrier(0): nothing crosses -- the order below IS the instruction stream)
        f32x16 acc = {0};
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (m < 8) {
#pragma unroll
                for (int r = 2 * m; r < 2 * m + 2; ++r) {
                    pr = __builtin_fmaf(-dim[r], w2i[r], pr);
                    pi = __builtin_fmaf(dim[r], w2r[r], pi);
                }
            } else if (m == 8) {
                const float nre = __builtin_fmaf(sre, tp.x, __builtin_fmaf(-sim, tp.y, __builtin_fmaf(sre, tp.z, -sim * tp.w))) + pr;
                const float nim = __builtin_fmaf(sre, tp.y, __builtin_fmaf(sim, tp.x, __builtin_fmaf(sre, tp.w, sim * tp.z))) + pi;
                sre = nre; sim = nim;
            } else if (m == 9) {
                if (unitEnded) finish(uMul - 1);
            }
            asm volatile("" : "+v"(pr), "+v"(pi), "+v"(sre), "+v"(sim));          // (pins this step's vector work between its neighbours' ...
#ifndef SGZ_RES_EXP_NO_MFMA
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[TP[m >> 1]][m & 1], Wr[TQ[m >> 1] * 2 + (m & 1)], acc, 0, 0, 0);
#else                                                                             // (platform experiment, results meaningless: no matrix instructions)
            acc[m] += float(A[TP[m >> 1]][m & 1][0]) * float(Wr[TQ[m >> 1] * 2 + (m & 1)][0]);
#endif
            asm volatile("" : "+v"(acc));                                         //  ... and the product behind it: both are pure values otherwise)
            SGZ_RES_SCHED_BARRIER();
        }
        dre = acc;
        // ---- phase B: the imaginary block sums of tile T  |  tile T + 1 parked, tile T + 2 requested, the real block sums folded
        f32x16 acc2 = {0};
        pr = 0.f; pi = 0.f;
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (m < PER) {                                           // one sample of tile T + 1 split and parked, its register refilled
                const int k = m;
                const float x = SINGLE ? nl[k] : __builtin_fmaf(crHeld, nr[k], clHeld * nl[k]);
                const float r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
                const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
                const uint32_t o = slot + uint32_t(k) * (TH / 32) * ROW;
                xs[buf ^ 1][0][o] = XsWord(__float_as_uint(x) >> 16); xs[buf ^ 1][1][o] = XsWord(__float_as_uint(r1) >> 16); xs[buf ^ 1][2][o] = XsWord(__float_as_uint(r2) >> 16);
                nl[k] = pl[tid + TH * k]; if constexpr (!SINGLE) nr[k] = pr2[tid + TH * k];
            } else {
#pragma unroll
                for (int r = 2 * (m - 4); r < 2 * (m - 4) + 2; ++r) {
                    pr = __builtin_fmaf(dre[r], w2r[r], pr);
                    pi = __builtin_fmaf(dre[r], w2i[r], pi);
                }
            }
            asm volatile("" : "+v"(pr), "+v"(pi));
#ifndef SGZ_RES_EXP_NO_MFMA
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[TP[m >> 1]][m & 1], Wi[TQ[m >> 1] * 2 + (m & 1)], acc2, 0, 0, 0);
#else
            acc2[m] += float(A[TP[m >> 1]][m & 1][1]) * float(Wi[TQ[m >> 1] * 2 + (m & 1)][1]);
#endif
            asm volatile("" : "+v"(acc2));
            SGZ_RES_SCHED_BARRIER();
        }
        dim = acc2;
        clHeld = clNext; crHeld = crNext;
        if (++tMul == tiles) { tMul = 0; ++uMul; }
    }
    // the last tile's imaginary block sums, and the last unit out
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        pr = __builtin_fmaf(-dim[r], w2i[r], pr);
        pi = __builtin_fmaf(dim[r], w2r[r], pi);
    }
    {
        const float nre = __builtin_fmaf(sre, tp.x, __builtin_fmaf(-sim, tp.y, __builtin_fmaf(sre, tp.z, -sim * tp.w))) + pr;
        const float nim = __builtin_fmaf(sre, tp.y, __builtin_fmaf(sim, tp.x, __builtin_fmaf(sre, tp.w, sim * tp.z))) + pi;
        sre = nre; sim = nim;
    }
    if (nT) finish(u1 - 1);
}

__device__ __forceinline__ void cmulD(double &re, double &im, double cr, double ci)
{
    const double nr = re * cr - im * ci, ni = re * ci + im * cr;
    re = nr; im = ni;
}

// chains the frames: s_f = c^hop s_{f-1} + local_f, in place (local_f becomes the state after frame f), and leaves the last state for the
// next call.  One thread per (pair, signal, vector, axis point): the only sequential part of a render, `frames` dependent complex
// multiply-adds per thread; the loads do not depend on the chain and run eight frames ahead.
template <int V>
__global__ __launch_bounds__(kResBlock) void resonatorChainKernel(ResParams prm)
{
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t unit = blockIdx.y;                              // (pair, signal, vector)
    const uint32_t v = unit % uint32_t(V), sg = (unit / uint32_t(V)) % uint32_t(prm.signals), pair = unit / (uint32_t(V) * uint32_t(prm.signals));
    const float4 c = prm.cpow[size_t(v) * prm.P + i];              // c^hop = hi + lo (plan.cpp): the low word keeps the chain from drifting off the sample-by-sample recurrence
    const size_t stride = size_t(prm.C) * size_t(prm.signals) * V * prm.P;      // one frame of `local`
    float2 *loc = prm.local + ((size_t(pair) * size_t(prm.signals) + sg) * V + v) * prm.P + i;
    float2 s = loc[0];                                             // frame 0 already continued from the carried state ...
    long f = 1;
    if (prm.allFromRest) {                                         // ... or it started from rest like the others: the chain starts at the carried state
        s = prm.state[(size_t(pair) * 2 + sg) * V * prm.P + size_t(v) * prm.P + i];
        f = 0;
    }
    // eight frames are multiplied while the next eight are in flight: a step is four dependent operations, a load a memory round trip
    constexpr int D = 8;
    float2 cur[D], nxt[D];
#pragma unroll
    for (int k = 0; k < D; ++k) cur[k] = f + k < prm.frames ? loc[size_t(f + k) * stride] : float2{0.f, 0.f};
    for (; f < prm.frames; f += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) nxt[k] = f + D + k < prm.frames ? loc[size_t(f + D + k) * stride] : float2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (f + k < prm.frames) {
                const float nre = __builtin_fmaf(s.x, c.x, __builtin_fmaf(-s.y, c.y, __builtin_fmaf(s.x, c.z, -s.y * c.w))) + cur[k].x;
                const float nim = __builtin_fmaf(s.x, c.y, __builtin_fmaf(s.y, c.x, __builtin_fmaf(s.x, c.w, s.y * c.z))) + cur[k].y;
                s = float2{nre, nim};
                loc[size_t(f + k) * stride] = s;
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) cur[k] = nxt[k];
    }
    prm.state[(size_t(pair) * 2 + sg) * V * prm.P + size_t(v) * prm.P + i] = s;
}

// every frame's windowed state as the planes K_B reads: getWholeWindowedState + the RSNT branch of mapToLinearSpace (:1103-1133) + the
// magnitude mapAndTransformDFTFilters takes first (sqrt(re^2 + im^2), :1329-1331, :1361-1366).  One thread per (frame, pair, axis point).
// the frequency-domain window over a signal's V resonator states (getWholeWindowedState; oracle/resonator.c
// sgzo_resonator_windowed_state): centre first, then -m, +m outwards, times the gain; operation by operation, no contraction
template <int V>
__device__ __forceinline__ void resWindowed(const ResParams &prm, const float (&sre)[V], const float (&sim)[V], float gain, float &ore, float &oim)
{
#pragma clang fp contract(off)
    constexpr int K = (V + 1) / 2;
    float re = prm.weights[K - 1] * sre[K - 1], im = prm.weights[K - 1] * sim[K - 1];
#pragma unroll
    for (int m = 1; m < K; ++m) {
        re = re + prm.weights[K - 1 - m] * sre[K - 1 - m];
        im = im + prm.weights[K - 1 - m] * sim[K - 1 - m];
        re = re + prm.weights[K - 1 + m] * sre[K - 1 + m];
        im = im + prm.weights[K - 1 + m] * sim[K - 1 + m];
    }
    ore = re * gain; oim = im * gain;
}
// the RSNT branch of mapToLinearSpace (:1103-1133) + the magnitude mapAndTransformDFTFilters takes first (sqrt(re^2 + im^2), :1329-1331,
// :1361-1366): the planes K_B reads
__device__ __forceinline__ void resEmit(const ResParams &prm, const float (&ore)[2], const float (&oim)[2], float *out)
{
#pragma clang fp contract(off)
    if (prm.mode == SGZ_CH_PHASE) {                                       // :1111-1127
        const float sr = ore[0] + ore[1], si = oim[0] + oim[1];
        const float cancellation = sqrtf(sr * sr + si * si);
        const float mid = sqrtf(ore[0] * ore[0] + oim[0] * oim[0]) + sqrtf(ore[1] * ore[1] + oim[1] * oim[1]);
        out[0] = mid;
        out[prm.P] = 1.0f - (mid > 0 ? cancellation / mid : 0.0f);
    } else {
        out[0] = sqrtf(ore[0] * ore[0] + oim[0] * oim[0]);
        if (prm.sides == 2) out[prm.P] = sqrtf(ore[1] * ore[1] + oim[1] * oim[1]);
    }
}

// every frame's windowed state as the planes K_B reads.  One thread per (frame, pair, axis point).
template <int V>
__global__ __launch_bounds__(kResBlock) void resonatorWindowKernel(ResParams prm)
{
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t pair = blockIdx.y % prm.C;
    const long f = long(blockIdx.y / prm.C);
    const int S = prm.signals;
    const float gain = prm.gain[i];
    float ore[2] = {0.f, 0.f}, oim[2] = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s >= S) break;
        const float2 *loc = prm.local + ((size_t(f) * prm.C + pair) * size_t(S) + size_t(s)) * V * prm.P + i;
        float sre[V], sim[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { const float2 z = loc[size_t(v) * prm.P]; sre[v] = z.x; sim[v] = z.y; }
        resWindowed<V>(prm, sre, sim, gain, ore[s], oim[s]);
    }
    resEmit(prm, ore, oim, prm.mapped + (size_t(f) * prm.C + pair) * size_t(prm.sides) * prm.P + i);
}

// ---- chain and window of a launch whose frames all started from rest (the matrix forms), parallel over SEGMENTS of frames: the
// sequential chain (one thread per resonator walking every frame, each step behind a memory round trip: 40 us of a 0.38 ms render)
// becomes  (1) every segment's end state from rest, all segments at once;  (2) the state entering every segment, folded from the
// carried state and the ends of the segments in front (fp64, pole^(hop len) by squaring the hi + lo pole^hop -- the sharded render's
// fold, sharded.hip);  (3) per segment the chain over the segment's frames with the window and the magnitudes behind every step: no second pass
// over the states, which are written back only when a caller needs them (skipWindow: the sharded render adds its carry first).
template <int V>
__global__ __launch_bounds__(kResBlock) void resonatorSegmentKernel(ResParams prm)
{
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t unit = blockIdx.y, g = blockIdx.z;               // (pair, signal, vector); segment (the last one's end is nobody's carry)
    const uint32_t v = unit % uint32_t(V);
    const float4 c = prm.cpow[size_t(v) * prm.P + i];
    const size_t stride = size_t(prm.C) * size_t(prm.signals) * V * prm.P;
    const long f0 = long(g) * prm.segLen, f1 = min(prm.frames, f0 + prm.segLen);
    const float2 *loc = prm.local + size_t(unit) * prm.P + i;
    float2 s{0.f, 0.f};
    constexpr int D = 4;
    for (long f = f0; f < f1; f += D) {
        float2 l[D];
#pragma unroll
        for (int k = 0; k < D; ++k) l[k] = f + k < f1 ? loc[size_t(f + k) * stride] : float2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D; ++k)
            if (f + k < f1) {
                const float nre = __builtin_fmaf(s.x, c.x, __builtin_fmaf(-s.y, c.y, __builtin_fmaf(s.x, c.z, -s.y * c.w))) + l[k].x;
                const float nim = __builtin_fmaf(s.x, c.y, __builtin_fmaf(s.y, c.x, __builtin_fmaf(s.x, c.w, s.y * c.z))) + l[k].y;
                s = float2{nre, nim};
            }
    }
    prm.segEnd[(size_t(g) * gridDim.y + unit) * prm.P + i] = s;
}

// the state entering every segment but the first, in place of the end state of the segment in front of it:
//     enter_{q+1} = pole^(hop segLen) enter_q + end_q,      enter_0 = the carried state,
// fp64, pole^(hop segLen) by squaring the hi + lo pole^hop.  One thread per resonator; the ends are all requested before the walk.
template <int V>
__global__ __launch_bounds__(kResBlock) void resonatorSegmentFoldKernel(ResParams prm, uint32_t ends)
{
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t unit = blockIdx.y;                              // (pair, signal, vector)
    const uint32_t v = unit % uint32_t(V), sg = (unit / uint32_t(V)) % uint32_t(prm.signals), pair = unit / (uint32_t(V) * uint32_t(prm.signals));
    const float4 c = prm.cpow[size_t(v) * prm.P + i];
    float2 e[kResSegments - 1];
#pragma unroll
    for (int q = 0; q < kResSegments - 1; ++q) e[q] = prm.segEnd[(size_t(min(uint32_t(q), ends - 1u)) * gridDim.y + unit) * prm.P + i];     // (no branch: a clamped index)
#pragma unroll
    for (int q = 0; q < kResSegments - 1; ++q) asm volatile("" ::"v"(e[q].x), "v"(e[q].y));      // (all in flight before the walk: it is one round trip, not `ends`)
    double pr = 1.0, pi = 0.0, br = double(c.x) + double(c.z), bi = double(c.y) + double(c.w);
    for (long x = prm.segLen; x > 0; x >>= 1) {
        if (x & 1) cmulD(pr, pi, br, bi);
        cmulD(br, bi, br, bi);
    }
    const float2 s0 = prm.state[(size_t(pair) * 2 + sg) * V * prm.P + size_t(v) * prm.P + i];
    // ... and a snapshot of the carried state for the FIRST segment, in the one slot no end state uses (index `ends`).  Until round 6 the
    // chain kernel's first-segment workgroups read prm.state themselves while its last-segment workgroups write the new state there -- in
    // the same launch.  On a device of its own every workgroup of that small launch starts at once and the read always came first; with
    // other processes time-slicing the device (pytest -n 4) a first-segment workgroup could start after a last-segment one had finished and
    // chain frames 0 .. from the wrong state: the "unreproduced wrong result" of round 5's fuzz campaign (seed 1005, case 12).
    prm.segEnd[(size_t(ends) * gridDim.y + unit) * prm.P + i] = s0;
    double sr = double(s0.x), si = double(s0.y);
#pragma unroll
    for (int q = 0; q < kResSegments - 1; ++q) {
        if (uint32_t(q) < ends) {                                  // (no break: e[] is indexed by constants and stays in registers)
            cmulD(sr, si, pr, pi);
            sr += double(e[q].x); si += double(e[q].y);
            prm.segEnd[(size_t(q) * gridDim.y + unit) * prm.P + i] = float2{float(sr), float(si)};
            sr = double(float(sr)); si = double(float(si));       // (the segment starts from the fp32 state it is handed)
        }
    }
}

template <int V>
__global__ __launch_bounds__(kResBlock) void resonatorChainWindowKernel(ResParams prm)
{
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t pair = blockIdx.y, g = blockIdx.z;
    const int S = prm.signals;
    const uint32_t units = prm.C * uint32_t(S) * uint32_t(V);
    const long f0 = long(g) * prm.segLen, f1 = min(prm.frames, f0 + prm.segLen);
    const float gain = prm.gain[i];
    float4 c[V];
    float sre[2][V], sim[2][V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        c[v] = prm.cpow[size_t(v) * prm.P + i];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s >= S) break;
            const size_t unit = (size_t(pair) * size_t(S) + size_t(s)) * V + v;
            // the state entering this segment: the carried one, or what resonatorSegmentFoldKernel left in the slot of the segment in front
            // (one segment: this workgroup reads the carried state and is the only one that writes it, behind its last frame.  Several:
            // the first segment takes the fold kernel's snapshot -- the last segment's workgroups overwrite prm.state in this same launch)
            const float2 s0 = gridDim.z == 1 ? prm.state[(size_t(pair) * 2 + s) * V * prm.P + size_t(v) * prm.P + i]
                                             : prm.segEnd[(size_t(g == 0 ? gridDim.z - 1 : g - 1) * units + unit) * prm.P + i];
            sre[s][v] = s0.x; sim[s][v] = s0.y;
        }
    }
    const size_t stride = size_t(units) * prm.P;
    float2 *loc = prm.local + size_t(pair) * size_t(S) * V * prm.P + i;
    constexpr int D = V <= 3 ? 2 : 1;                                // frames requested ahead (2 S V values each; 4 measured no faster)
    for (long f = f0; f < f1; f += D) {
        float2 l[D][2][V];
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int v = 0; v < V; ++v)
                    l[k][s][v] = (s < S && f + k < f1) ? loc[size_t(f + k) * stride + (size_t(s) * V + v) * prm.P] : float2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D; ++k) {
            if (f + k >= f1) break;
            float ore[2] = {0.f, 0.f}, oim[2] = {0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (s >= S) break;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const float nre = __builtin_fmaf(sre[s][v], c[v].x, __builtin_fmaf(-sim[s][v], c[v].y, __builtin_fmaf(sre[s][v], c[v].z, -sim[s][v] * c[v].w))) + l[k][s][v].x;
                    const float nim = __builtin_fmaf(sre[s][v], c[v].y, __builtin_fmaf(sim[s][v], c[v].x, __builtin_fmaf(sre[s][v], c[v].w, sim[s][v] * c[v].z))) + l[k][s][v].y;
                    sre[s][v] = nre; sim[s][v] = nim;
                    if (prm.skipWindow) loc[size_t(f + k) * stride + (size_t(s) * V + v) * prm.P] = float2{nre, nim};
                }
                if (!prm.skipWindow) resWindowed<V>(prm, sre[s], sim[s], gain, ore[s], oim[s]);
            }
            if (!prm.skipWindow) resEmit(prm, ore, oim, prm.mapped + (size_t(f + k) * prm.C + pair) * size_t(prm.sides) * prm.P + i);
        }
    }
    if (g + 1 == gridDim.z) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s >= S) break;
#pragma unroll
            for (int v = 0; v < V; ++v) prm.state[(size_t(pair) * 2 + s) * V * prm.P + size_t(v) * prm.P + i] = float2{sre[s][v], sim[s][v]};
        }
    }
}

template <int V>
hipError_t launchWindow(const ResParams &prm, hipStream_t stream);

template <int V>
hipError_t launchV(const ResParams &prm0, hipStream_t stream)
{
    ResParams prm = prm0;
    const unsigned tiles = (prm.P + kResBlock - 1) / kResBlock;
    // a launch of several frames at a hop the plan has the weights for (a multiple of 1024): EVERY frame from rest on the matrix cores,
    // the chain starts at the carried state.  (Until round 4 frame 0 continued the carried state sample by sample on the vector ALUs
    // beside the matrix kernel: 8 workgroups walking `hop` dependent steps, 0.25-0.4 ms -- longer than the matrix kernel is now.)
    // A one-frame launch -- the real-time case -- is the reference's recurrence step for step.
    const bool matrix = prm.w1 && prm.frames > 1 && prm.hop % 1024u == 0;
    prm.allFromRest = matrix;
    // grid.y is limited to 65535: long renders go in slabs of frames
    const long perSlab = std::max<long>(1, long(65535u / (prm.C * uint32_t(prm.signals))));
    for (long f0 = 0; f0 < prm.frames; f0 += perSlab) {
        ResParams q = prm;
        const long nf = std::min(perSlab, prm.frames - f0);
        q.frames = nf;
        q.planar = prm.planar + size_t(f0) * prm.hop;
        q.local = prm.local + size_t(f0) * prm.C * size_t(prm.signals) * V * prm.P;
        q.firstContinues = f0 == 0;
        if (!matrix) hipLaunchKernelGGL(resonateKernel<V>, dim3(tiles, unsigned(nf * prm.C * prm.signals)), dim3(kResBlock), 0, stream, q);
        else {
            const unsigned groups = (prm.P + 31) / 32 * unsigned(V);
            constexpr int WAVES = SGZ_RES_WAVES;
            const unsigned units = unsigned(nf * prm.C * prm.signals);
            // the bf16 form walks kResUnitsPerWg units per workgroup, fewer when that would leave compute units without work
            unsigned upw = kResUnitsPerWg;
            while (upw > 1 && size_t((groups + WAVES - 1) / WAVES) * ((units + upw - 1) / upw) < 4096) upw /= 2;
            if (prm.matrixForm == 2) hipLaunchKernelGGL(resonateMfmaKernel<WAVES>, dim3((groups + WAVES - 1) / WAVES, units), dim3(64 * WAVES), 0, stream, q, V);
            else if (prm.mode != SGZ_CH_MERGE && prm.mode != SGZ_CH_SIDE && prm.mode != SGZ_CH_MIDSIDE)
                hipLaunchKernelGGL((resonateMfmaBf16Kernel<WAVES, true>), dim3((groups + WAVES - 1) / WAVES, (units + upw - 1) / upw), dim3(64 * WAVES), 0, stream, q, V, units, upw);
            else hipLaunchKernelGGL((resonateMfmaBf16Kernel<WAVES, false>), dim3((groups + WAVES - 1) / WAVES, (units + upw - 1) / upw), dim3(64 * WAVES), 0, stream, q, V, units, upw);
        }
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    if (matrix && prm.segEnd) {
        // (the carried state is read by the fold kernel only -- which leaves the first segment a snapshot of it -- and written by the
        //  chain kernel's last segment: the read and the write are in different launches of one stream)
        const long segs = std::max<long>(1, std::min<long>(kResSegments, prm.frames / 8));
        prm.segLen = (prm.frames + segs - 1) / segs;
        const unsigned G = unsigned((prm.frames + prm.segLen - 1) / prm.segLen);
        if (G > 1) {
            hipLaunchKernelGGL(resonatorSegmentKernel<V>, dim3(tiles, prm.C * unsigned(prm.signals) * unsigned(V), G - 1), dim3(kResBlock), 0, stream, prm);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
            hipLaunchKernelGGL(resonatorSegmentFoldKernel<V>, dim3(tiles, prm.C * unsigned(prm.signals) * unsigned(V)), dim3(kResBlock), 0, stream, prm, G - 1);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(resonatorChainWindowKernel<V>, dim3(tiles, prm.C, G), dim3(kResBlock), 0, stream, prm);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(resonatorChainKernel<V>, dim3(tiles, prm.C * unsigned(prm.signals) * unsigned(V)), dim3(kResBlock), 0, stream, prm);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    if (prm.skipWindow) return hipSuccess;
    return launchWindow<V>(prm, stream);
}

template <int V>
hipError_t launchWindow(const ResParams &prm, hipStream_t stream)
{
    const unsigned tiles = (prm.P + kResBlock - 1) / kResBlock;
    for (long f0 = 0; f0 < prm.frames; f0 += long(65535u / prm.C)) {
        ResParams q = prm;
        const long nf = std::min<long>(long(65535u / prm.C), prm.frames - f0);
        q.local = prm.local + size_t(f0) * prm.C * size_t(prm.signals) * V * prm.P;
        q.mapped = prm.mapped + size_t(f0) * prm.C * size_t(prm.sides) * prm.P;
        hipLaunchKernelGGL(resonatorWindowKernel<V>, dim3(tiles, unsigned(nf) * prm.C), dim3(kResBlock), 0, stream, q);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

struct FoldFrames { long long n[64]; };

// the state entering rank `rank`: s <- pole^(frames_q hop) s + end_q over the ranks in front, fp64.  One thread per (pair, signal slot,
// vector, axis point) = one entry of the plan's state layout [C][2][V][P].
__global__ __launch_bounds__(kResBlock) void resonatorFoldKernel(ResParams prm, const float2 *allEnd, FoldFrames fr, uint32_t rank, float2 *carry)
{
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t unit = blockIdx.y;                              // (pair, slot, vector)
    const uint32_t v = unit % uint32_t(prm.V);
    const size_t at = size_t(unit) * prm.P + i, perRank = size_t(prm.C) * 2 * size_t(prm.V) * prm.P;
    const float4 c = prm.cpow[size_t(v) * prm.P + i];              // pole^hop = hi + lo
    const double hr = double(c.x) + double(c.z), hi = double(c.y) + double(c.w);
    double sr = 0.0, si = 0.0;
    for (uint32_t q = 0; q < rank; ++q) {
        // pole^(frames_q hop) by squaring
        double pr = 1.0, pi = 0.0, br = hr, bi = hi;
        for (long long e = fr.n[q]; e > 0; e >>= 1) {
            if (e & 1) cmulD(pr, pi, br, bi);
            cmulD(br, bi, br, bi);
        }
        cmulD(sr, si, pr, pi);
        const float2 eq = allEnd[size_t(q) * perRank + at];
        sr += double(eq.x); si += double(eq.y);
    }
    carry[at] = float2{float(sr), float(si)};
}

// local_f += pole^((f + 1) hop) carry, f = 0 .. frames - 1 (fp64 walk); the plan's state (the end state from rest) likewise
template <int V>
__global__ __launch_bounds__(kResBlock) void resonatorCarryKernel(ResParams prm, const float2 *carry)
{
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t unit = blockIdx.y;                              // (pair, signal, vector)
    const uint32_t v = unit % uint32_t(V), sg = (unit / uint32_t(V)) % uint32_t(prm.signals), pair = unit / (uint32_t(V) * uint32_t(prm.signals));
    const float4 c = prm.cpow[size_t(v) * prm.P + i];
    const double hr = double(c.x) + double(c.z), hi = double(c.y) + double(c.w);
    const size_t stAt = (size_t(pair) * 2 + sg) * V * prm.P + size_t(v) * prm.P + i;
    const float2 c0 = carry[stAt];
    double tr = double(c0.x), ti = double(c0.y);
    const size_t stride = size_t(prm.C) * size_t(prm.signals) * V * prm.P;
    float2 *loc = prm.local + ((size_t(pair) * size_t(prm.signals) + sg) * V + v) * prm.P + i;
    for (long f = 0; f < prm.frames; ++f) {
        cmulD(tr, ti, hr, hi);
        float2 l = loc[size_t(f) * stride];
        l.x = float(double(l.x) + tr); l.y = float(double(l.y) + ti);
        loc[size_t(f) * stride] = l;
    }
    float2 s = prm.state[stAt];
    s.x = float(double(s.x) + tr); s.y = float(double(s.y) + ti);
    prm.state[stAt] = s;
}

template <int V>
hipError_t launchCarryV(const ResParams &prm, const float2 *carry, hipStream_t stream)
{
    const unsigned tiles = (prm.P + kResBlock - 1) / kResBlock;
    if (carry) {
        hipLaunchKernelGGL(resonatorCarryKernel<V>, dim3(tiles, prm.C * unsigned(prm.signals) * unsigned(V)), dim3(kResBlock), 0, stream, prm, carry);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return launchWindow<V>(prm, stream);
}

}  // namespace

hipError_t launchResonator(const ResParams &prm, hipStream_t stream)
{
    switch (prm.V) {
    case 1: return launchV<1>(prm, stream);
    case 3: return launchV<3>(prm, stream);
    case 5: return launchV<5>(prm, stream);
    case 7: return launchV<7>(prm, stream);
    case 9: return launchV<9>(prm, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launchResonatorFold(const ResParams &prm, const float2 *allEnd, const long long *framesPerRank, uint32_t world, uint32_t rank,
                               float2 *carry, hipStream_t stream)
{
    if (world > 64 || rank >= world) return hipErrorInvalidValue;
    FoldFrames fr{};
    for (uint32_t q = 0; q < world; ++q) fr.n[q] = framesPerRank[q];
    const unsigned tiles = (prm.P + kResBlock - 1) / kResBlock;
    hipLaunchKernelGGL(resonatorFoldKernel, dim3(tiles, prm.C * 2u * unsigned(prm.V)), dim3(kResBlock), 0, stream, prm, allEnd, fr, rank, carry);
    return hipGetLastError();
}

hipError_t launchResonatorCarry(const ResParams &prm, const float2 *carry, hipStream_t stream)
{
    switch (prm.V) {
    case 1: return launchCarryV<1>(prm, carry, stream);
    case 3: return launchCarryV<3>(prm, carry, stream);
    case 5: return launchCarryV<5>(prm, carry, stream);
    case 7: return launchCarryV<7>(prm, carry, stream);
    case 9: return launchCarryV<9>(prm, carry, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace sgz
