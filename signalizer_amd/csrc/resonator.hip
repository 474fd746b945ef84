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
// boundaries: frame f's workgroups run the recurrence over that frame's `hop` samples -- frame 0 continuing from the carried state
// (sequential semantics: a one-frame launch, the real-time case, is the reference's recurrence step for step), frames f > 0 from
// rest -- and a fold kernel chains them:  s_f = c^hop s_{f-1} + local_f,  c^hop evaluated in double from the fp32 pole on the host.
// Every thread owns one axis point (all V vectors of it: V independent dependency chains); the samples are uniform across a
// workgroup and are staged through LDS 256 at a time (one coalesced load + channel mix per thread, then broadcast reads).
// Against the sequential fp32 recurrence the chained result differs by the rounding of one complex product per frame: ~1e-7 of the
// state.  Bytes: 8 hop per (frame, pair) in, 8 V P per (frame, signal) through HBM between the two kernels: the path is
// VALU-bound (per sample and axis point 7 V fp32 operations without contraction in the continuing frame, ~2.4 V fused ones in the
// others, which take the samples eight at a time against the pole's powers).
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

// chains the frames (s_f = c^hop s_{f-1} + local_f), leaves the last state for the next call and writes every frame's windowed state
// as the planes K_B reads: getWholeWindowedState + the RSNT branch of mapToLinearSpace (:1103-1133) + the magnitude
// mapAndTransformDFTFilters takes first (sqrt(re^2 + im^2), :1329-1331, :1361-1366)
template <int V>
__global__ __launch_bounds__(kResBlock) void resonatorFoldKernel(ResParams prm)
{
#pragma clang fp contract(off)
    const uint32_t i = blockIdx.x * kResBlock + threadIdx.x;
    if (i >= prm.P) return;
    const uint32_t pair = blockIdx.y;
    constexpr int K = (V + 1) / 2;
    const int S = prm.signals;
    float pr[V], pi[V], qr[V], qi[V], w[V];
    float sre[2][V], sim[2][V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const float4 c = prm.cpow[size_t(v) * prm.P + i];
        pr[v] = c.x; pi[v] = c.y; qr[v] = c.z; qi[v] = c.w; w[v] = prm.weights[v];
#pragma unroll
        for (int s = 0; s < 2; ++s) { sre[s][v] = 0.f; sim[s][v] = 0.f; }
    }
    const float gain = prm.gain[i];
    for (long f = 0; f < prm.frames; ++f) {
        float ore[2] = {0.f, 0.f}, oim[2] = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s >= S) break;
            const float2 *loc = prm.local + ((size_t(f) * prm.C + pair) * size_t(S) + size_t(s)) * V * prm.P + i;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float2 l = loc[size_t(v) * prm.P];
                if (f == 0) { sre[s][v] = l.x; sim[s][v] = l.y; }            // frame 0 already continued from the carried state
                else {
                    // c^hop = hi + lo (plan.cpp): the low word keeps the chain from drifting off the sample-by-sample recurrence
                    const float nre = __builtin_fmaf(sre[s][v], pr[v], __builtin_fmaf(-sim[s][v], pi[v], __builtin_fmaf(sre[s][v], qr[v], -sim[s][v] * qi[v]))) + l.x;
                    const float nim = __builtin_fmaf(sre[s][v], pi[v], __builtin_fmaf(sim[s][v], pr[v], __builtin_fmaf(sre[s][v], qi[v], sim[s][v] * qr[v]))) + l.y;
                    sre[s][v] = nre; sim[s][v] = nim;
                }
            }
            // frequency-domain window: centre first, then -m, +m outwards (oracle/resonator.c sgzo_resonator_windowed_state)
            float re = w[K - 1] * sre[s][K - 1], im = w[K - 1] * sim[s][K - 1];
#pragma unroll
            for (int m = 1; m < K; ++m) {
                re = re + w[K - 1 - m] * sre[s][K - 1 - m];
                im = im + w[K - 1 - m] * sim[s][K - 1 - m];
                re = re + w[K - 1 + m] * sre[s][K - 1 + m];
                im = im + w[K - 1 + m] * sim[s][K - 1 + m];
            }
            ore[s] = re * gain; oim[s] = im * gain;
        }
        float *out = prm.mapped + (size_t(f) * prm.C + pair) * size_t(prm.sides) * prm.P + i;
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
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s >= S) break;
#pragma unroll
        for (int v = 0; v < V; ++v) prm.state[(size_t(pair) * 2 + size_t(s)) * V * prm.P + size_t(v) * prm.P + i] = float2{sre[s][v], sim[s][v]};
    }
}

template <int V>
hipError_t launchV(const ResParams &prm, hipStream_t stream)
{
    const unsigned tiles = (prm.P + kResBlock - 1) / kResBlock;
    // grid.y is limited to 65535: long renders go in slabs of frames (the kernel reads the frame from blockIdx.y plus the slab's base)
    const long perSlab = std::max<long>(1, long(65535u / (prm.C * uint32_t(prm.signals))));
    for (long f0 = 0; f0 < prm.frames; f0 += perSlab) {
        ResParams q = prm;
        const long nf = std::min(perSlab, prm.frames - f0);
        q.frames = nf;
        q.planar = prm.planar + size_t(f0) * prm.hop;
        q.local = prm.local + size_t(f0) * prm.C * size_t(prm.signals) * V * prm.P;
        q.firstContinues = f0 == 0;
        hipLaunchKernelGGL(resonateKernel<V>, dim3(tiles, unsigned(nf * prm.C * prm.signals)), dim3(kResBlock), 0, stream, q);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(resonatorFoldKernel<V>, dim3(tiles, prm.C), dim3(kResBlock), 0, stream, prm);
    return hipGetLastError();
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

}  // namespace sgz
