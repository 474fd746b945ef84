// spectrum_fft.hip -- K_A: fused  window x audio -> N-point complex FFT -> two-for-one split -> |X| ->
// pixel mapping (interpolate / arg-max) for one (frame, stereo pair) per workgroup.  gfx950 only.
//
// Replaces, per frame: TransformPair::prepareTransform (Source/Spectrum/TransformDSP.inl:39-231),
// doTransform (:487-502, cpl::dsp::UniFFT forward), and mapToLinearSpace (:506-1102) up to csp[].
//
// Structure (N = R^3, R = 32 for N = 32768, R = 16 for N = 4096; T = R^2 threads, R points per thread,
// all butterflies in VGPRs, LDS only for the two digit transposes and the k <-> N-k mirror):
//   pass 1  thread t        : R-point DIF over x[t + T j]  (coalesced strided HBM/L2 loads, window fused),
//                             times W_N^{t q}            -> exchange 1 (workgroup-wide, re then im)
//   pass 2  thread (q,t2)   : R-point DIF over j2,  times W_T^{t2 q2}  -> exchange 2 (inside R-lane groups)
//   pass 3  thread (q,q2)   : R-point DIF over t2  -> X[q + R q2 + T m3]
//   mirror  : Z[k], Z[N-k] meet through LDS (re then im) -> M[k] = |X1[k]|, M[N-k] = |X2[k]|  (csf of the
//             reference after :858-869), kept in LDS in a bank-padded natural order
//   mapping : one thread per (side, pixel) record (plan.cpp) -> csp magnitude, written to HBM (8 KB / frame)
// HBM/L2 traffic per frame-pair: 2*W*4 B audio + W*4 B window + N*8 B twiddles (L2 resident tables) in,
// sides*P*4 B out.  No MFMA: the path is bandwidth/LDS bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace sgz {

// ---- compile-time twiddles W_32^j = cos(2 pi j/32) - i sin(2 pi j/32), j = 0..16 ---------------------
__host__ __device__ constexpr float cos32(int j)
{
    constexpr float v[17] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                             0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                             -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                             -0.92387953251128675613f, -0.98078528040323044913f, -1.0f};
    return v[j];
}
__host__ __device__ constexpr float sin32(int j) { return j <= 8 ? cos32(8 - j) : cos32(j - 8); }

__host__ __device__ constexpr int brev(int x, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
    return r;
}

// In-register radix-2 DIF over LEN elements starting at BASE; result is in bit-reversed order.
template <int R, int LEN, int BASE>
__device__ __forceinline__ void dif(float (&re)[R], float (&im)[R])
{
    constexpr int H = LEN / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int a = BASE + i, b = BASE + i + H;
        const float ar = re[a], ai = im[a], br = re[b], bi = im[b];
        re[a] = ar + br;
        im[a] = ai + bi;
        const float dr = ar - br, di = ai - bi;
        const int j = i * (32 / LEN);
        if (j == 0) { re[b] = dr; im[b] = di; }
        else if (j == 8) { re[b] = di; im[b] = -dr; }
        else {
            const float c = cos32(j), s = sin32(j);
            re[b] = dr * c + di * s;
            im[b] = di * c - dr * s;
        }
    }
    if constexpr (LEN > 2) {
        dif<R, H, BASE>(re, im);
        dif<R, H, BASE + H>(re, im);
    }
}

// Buffer-resource (SRSRC) loads: one wave-uniform descriptor + a 32-bit per-lane offset + a scalar offset,
// so the 3R strided loads of a thread need no 64-bit address VGPRs, and reads past `bytes` return 0
// (that is the zero padding of prepareTransform, TransformDSP.inl:220-223, for W < N).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t makeRsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, int(bytes), 0x00020000);
}
__device__ __forceinline__ float bufLoad(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// NOTE: __builtin_amdgcn_raw_buffer_load_b64/_b128 are mis-lowered to a single buffer_load_dword by this
// ROCm 7.2 hipcc (verified in the ISA), so a complex twiddle is fetched as two dword loads.
__device__ __forceinline__ float2 bufLoad2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const float x = bufLoad(r, voff, soff);
    const float y = bufLoad(r, voff + 4, soff);
    return make_float2(x, y);
}

// Pixel mapping of mapToLinearSpace (TransformDSP.inl:565-639, :871-985) on the csf magnitudes held in LDS
// (bank-padded natural order).  Every operation rounds exactly like the reference's scalar fp32 code:
// contraction is off in this function (NB: hip's __fmul_rn/__fadd_rn are plain * and + and would be fused,
// and __fsqrt_rn is the approximate native sqrt -- neither is used here).
template <int LR>
__device__ __forceinline__ void mapPixels(const StftParams &prm, const float *lds, int tid, long task)
{
#pragma clang fp contract(off)
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const int total = int(prm.sides * prm.P);
    float *out = prm.mapped + size_t(task) * total;
    for (int idx = tid; idx < total; idx += T) {
        const PixelRec rec = prm.recs[idx];
        const int side = idx >= int(prm.P) ? 1 : 0;
        float val;
        if (rec.kind == 0) {
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
        // mapAndTransformDFTFilters: magnitude = sqrt(re*re + im*im), im == 0 (TransformDSP.inl:1331,:1365)
        const float sq = val * val + 0.f;
        out[idx] = __builtin_sqrtf(sq);                               // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    }
}

// One workgroup = one (frame, pair).  LR = log2(R).
template <int LR>
__global__ void __launch_bounds__(1 << (2 * LR))
stftMapKernel(const StftParams prm)
{
    constexpr int R = 1 << LR;
    constexpr int T = R * R;
    constexpr int N = R * T;
    constexpr int PADSTRIDE = T + (T >> LR);          // padded distance between k and k + T
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float2 *ldsc = reinterpret_cast<float2 *>(lds);

    const int tid = threadIdx.x;
    const long tasks = prm.frames * long(prm.C);

    for (long task = blockIdx.x; task < tasks; task += gridDim.x) {
        const long frame = task / prm.C;
        const int pair = int(task - frame * prm.C);
        float re[R], im[R];

        // ------------------------------------------------------------------ pass 1: load + window + DIF
        if (prm.binsIn == nullptr) {
            const float *L = prm.planar + size_t(2 * pair) * prm.chStride + size_t(frame) * prm.hop;
            const __amdgpu_buffer_rsrc_t rsL = makeRsrc(L, prm.W * 4u);
            const __amdgpu_buffer_rsrc_t rsR = makeRsrc(L + prm.chStride, prm.W * 4u);
            const __amdgpu_buffer_rsrc_t rsW = makeRsrc(prm.window, prm.W * 4u);
            const int mode = prm.mode;
            const int voff4 = tid * 4, voff8 = tid * 8;
            constexpr int LB = 8;                                   // loads in flight per batch (bounds VGPR liveness)
#pragma unroll
            for (int jb = 0; jb < R; jb += LB) {
#pragma unroll
                for (int j = jb; j < jb + LB; ++j) {
                    const float l = bufLoad(rsL, voff4, j * (T * 4));
                    const float r = bufLoad(rsR, voff4, j * (T * 4));
                    const float w = bufLoad(rsW, voff4, j * (T * 4));
                    float xr, xi;
                    switch (mode) {                                   // prepareTransform, TransformDSP.inl:59-216
                    case SGZ_CH_LEFT: xr = l * w; xi = 0.f; break;
                    case SGZ_CH_RIGHT: xr = r * w; xi = 0.f; break;
                    case SGZ_CH_MERGE: xr = (l + r) * w * 0.5f; xi = 0.f; break;
                    case SGZ_CH_SIDE: xr = (l - r) * w * 0.5f; xi = 0.f; break;
                    case SGZ_CH_MIDSIDE: xr = (l + r) * w * 0.5f; xi = (l - r) * w * 0.5f; break;
                    default: xr = l * w; xi = r * w; break;
                    }
                    re[j] = xr; im[j] = xi;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            dif<R, R, 0>(re, im);
            // twiddle W_N^{t q}
            __builtin_amdgcn_sched_barrier(0);
            const __amdgpu_buffer_rsrc_t rsT1 = makeRsrc(prm.tw1, uint32_t(R) * T * 8u);
#pragma unroll
            for (int qb = 0; qb < R; qb += LB) {
#pragma unroll
                for (int q = qb; q < qb + LB; ++q) {
                    if (q == 0) continue;
                    const float2 w = bufLoad2(rsT1, voff8, q * (T * 8));
                    const int i = brev(q, LR);
                    const float a = re[i], b = im[i];
                    re[i] = a * w.x - b * w.y;
                    im[i] = a * w.y + b * w.x;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---------------------------------------------------------- exchange 1 (workgroup; re then im)
            {
                const int q = tid >> LR, t2 = tid & (R - 1);
                __syncthreads();
#pragma unroll
                for (int qq = 0; qq < R; ++qq) lds[qq * T + tid] = re[brev(qq, LR)];
                __syncthreads();
#pragma unroll
                for (int j2 = 0; j2 < R; ++j2) re[j2] = lds[q * T + t2 + R * j2];
                __syncthreads();
#pragma unroll
                for (int qq = 0; qq < R; ++qq) lds[qq * T + tid] = im[brev(qq, LR)];
                __syncthreads();
#pragma unroll
                for (int j2 = 0; j2 < R; ++j2) im[j2] = lds[q * T + t2 + R * j2];
            }
            // ------------------------------------------------------------------ pass 2
            dif<R, R, 0>(re, im);
            {
                const int q = tid >> LR, t2 = tid & (R - 1);
                __builtin_amdgcn_sched_barrier(0);
                const __amdgpu_buffer_rsrc_t rsT2 = makeRsrc(prm.tw2, uint32_t(R) * R * 8u);
#pragma unroll
                for (int qb = 0; qb < R; qb += 8) {
#pragma unroll
                    for (int q2 = qb; q2 < qb + 8; ++q2) {
                        if (q2 == 0) continue;
                        const float2 w = bufLoad2(rsT2, t2 * 8, q2 * (R * 8));
                        const int i = brev(q2, LR);
                        const float a = re[i], b = im[i];
                        re[i] = a * w.x - b * w.y;
                        im[i] = a * w.y + b * w.x;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // -------------------------------------------------- exchange 2 (inside R-lane groups, re then im)
                const int base = q * (R * (R + 1));
                __syncthreads();
#pragma unroll
                for (int q2 = 0; q2 < R; ++q2) lds[base + q2 * (R + 1) + t2] = re[brev(q2, LR)];
                __syncthreads();
                // now this thread plays (q, q2 = t2)
#pragma unroll
                for (int j = 0; j < R; ++j) re[j] = lds[base + t2 * (R + 1) + j];
                __syncthreads();
#pragma unroll
                for (int q2 = 0; q2 < R; ++q2) lds[base + q2 * (R + 1) + t2] = im[brev(q2, LR)];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < R; ++j) im[j] = lds[base + t2 * (R + 1) + j];
            }
            // ------------------------------------------------------------------ pass 3
            dif<R, R, 0>(re, im);
            // thread (q, q2): X[c + T m3] at index brev(m3), c = q + R q2
            const int q = tid >> LR, q2 = tid & (R - 1);
            const int c = q + R * q2;
            const int ownBase = c + (c >> LR);                    // padded address of k = c
            const bool split = (prm.sides == 2);
            float dcRe = 0.f, dcIm = 0.f, nyRe = 0.f, nyIm = 0.f;
            if (c == 0) { dcRe = re[0]; dcIm = im[0]; nyRe = re[brev(R / 2, LR)]; nyIm = im[brev(R / 2, LR)]; }

            if (split) {
                // mirror partner N-k of k = c + T m3:  c != 0: (T-c) + T (R-1-m3);  c == 0: T (R-m3)
                // (for c == 0 the m3 = 0 slot degenerates to the csf[N] slot; thread 0 rewrites k = 0, N/2 below)
                const int cm = T - c;
                const int mirBase = (c != 0) ? (cm + (cm >> LR)) : PADSTRIDE;
                // ---- real parts
                __syncthreads();
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) lds[ownBase + m3 * PADSTRIDE] = re[brev(m3, LR)];
                __syncthreads();
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) {
                    const float mr = lds[mirBase + (R - 1 - m3) * PADSTRIDE];
                    const int i = brev(m3, LR);
                    // k < N/2 (m3 < R/2): u = own.re + mirror.re ; k > N/2: u = own.re - mirror.re
                    re[i] = (m3 < R / 2) ? (re[i] + mr) : (re[i] - mr);
                }
                // ---- imaginary parts
                __syncthreads();
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) lds[ownBase + m3 * PADSTRIDE] = im[brev(m3, LR)];
                __syncthreads();
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) {
                    const float mi = lds[mirBase + (R - 1 - m3) * PADSTRIDE];
                    const int i = brev(m3, LR);
                    im[i] = (m3 < R / 2) ? (im[i] - mi) : (im[i] + mi);
                }
                __syncthreads();
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) {
                    const int i = brev(m3, LR);
                    lds[ownBase + m3 * PADSTRIDE] = 0.5f * __builtin_amdgcn_sqrtf(re[i] * re[i] + im[i] * im[i]);
                }
                __syncthreads();
                if (tid == 0) {                                      // TransformDSP.inl:861-864
                    lds[N + (N >> LR)] = dcIm * 0.5f;                // csf[N]   = Im(csf[0]) * 0.5
                    lds[0] = dcRe * 0.5f;                            // csf[0]   = Re(csf[0]) * 0.5
                    lds[N / 2 + ((N / 2) >> LR)] = 0.5f * __fsqrt_rn(nyRe * nyRe + nyIm * nyIm);   // csf[N/2] *= 0.5
                    const int kq = N / 2 - 1;
                    lds[kq + (kq >> LR)] *= 0.5f;                    // csf[N/2-1] *= 0.5 (quirk Q3)
                }
            } else {
                // mono / complex modes: csf[k] = |Z[k]| (TransformDSP.inl:553-560, :993-1002)
                __syncthreads();
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) {
                    const int i = brev(m3, LR);
                    lds[ownBase + m3 * PADSTRIDE] = __builtin_amdgcn_sqrtf(re[i] * re[i] + im[i] * im[i]);
                }
                __syncthreads();
                if (tid == 0) {
                    lds[N + (N >> LR)] = 0.f;
                    lds[0] = 0.5f * __fsqrt_rn(dcRe * dcRe + dcIm * dcIm);
                    if (prm.mode != SGZ_CH_COMPLEX)
                        lds[N / 2 + ((N / 2) >> LR)] = 0.5f * __fsqrt_rn(nyRe * nyRe + nyIm * nyIm);
                }
            }
            __syncthreads();
        } else {
            // test path (sgz_stage_map_from_bins): csf magnitudes come from HBM
            const float *src = prm.binsIn + size_t(task) * (N + 1);
            __syncthreads();
            for (int k = tid; k <= N; k += T) lds[k + (k >> LR)] = src[k];
            __syncthreads();
        }

        if (prm.binsOut) {
            float *dst = prm.binsOut + size_t(task) * (N + 1);
            for (int k = tid; k <= N; k += T) dst[k] = lds[k + (k >> LR)];
        }

        // ---------------------------------------------------------------------- pixel mapping
        if (prm.mapped) mapPixels<LR>(prm, lds, tid, task);
    }
}

template <int LR>
static hipError_t launchStft(const StftParams &prm, int grid, hipStream_t stream)
{
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const size_t ldsBytes = (size_t(N) + (N >> LR) + 4) * sizeof(float);
    static bool attrSet = false;
    if (!attrSet) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&stftMapKernel<LR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    hipLaunchKernelGGL(stftMapKernel<LR>, dim3(grid), dim3(T), ldsBytes, stream, prm);
    return hipGetLastError();
}

hipError_t launchStftMap(const StftParams &prm, uint32_t N, int grid, hipStream_t stream)
{
    switch (N) {
    case 32768: return launchStft<5>(prm, grid, stream);
    case 4096: return launchStft<4>(prm, grid, stream);
    default: return hipErrorNotSupported;
    }
}

}  // namespace sgz
