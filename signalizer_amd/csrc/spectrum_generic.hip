// spectrum_generic.hip -- K_A for ANY power-of-two transform size (32 <= N, not only the fused 4096 / 32768
// kernel of spectrum_fft.hip): the same stages as separate HBM-resident kernels.  gfx950 only.
//
//   genericPrepare : prepareTransform (Source/Spectrum/TransformDSP.inl:39-231)   audio x window -> complex [tasks][N]
//   genericStage   : one radix-2 Stockham (autosort) pass of the forward DFT; log2 N launches, ping-pong in HBM
//                    (doTransform, :487-502 -- natural order, unnormalised)
//   genericBins    : two-for-one split + DC/Nyquist fix-ups + |.|  (:858-869 ; mono :553-560)  -> csf magnitudes [N+1]
//   genericMap     : pixel mapping from HBM-resident csf (:565-639, :871-985), same PixelRec table as the fused kernel
//
// This path moves ~(2 log2 N + 6) * N * 8 bytes per frame-pair through HBM/L2 instead of the fused kernel's
// 2*W*4: it exists for completeness (every window size the reference accepts works, e.g. BASELINE cfg5's
// N = 65536), not for speed; fusing N = 2 R^3 is listed under "next" in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace sgz {

__global__ void __launch_bounds__(256)
genericPrepare(const float *planar, size_t chStride, uint32_t hop, uint32_t W, uint32_t N, uint32_t C, uint32_t mode,
               const float *window, long task0, long ntasks, float2 *out)
{
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(ntasks) * N) return;
    const long t = long(gid / N);
    const uint32_t n = uint32_t(gid - size_t(t) * N);
    const long task = task0 + t;
    const long frame = task / C;
    const uint32_t pair = uint32_t(task - frame * C);
    float xr = 0.f, xi = 0.f;
    if (n < W) {
        const float *L = planar + size_t(2 * pair) * chStride + size_t(frame) * hop;
        const float l = L[n], r = L[chStride + n], w = window[n];
        switch (mode) {                                           // TransformDSP.inl:59-216
        case SGZ_CH_LEFT: xr = l * w; break;
        case SGZ_CH_RIGHT: xr = r * w; break;
        case SGZ_CH_MERGE: xr = (l + r) * w * 0.5f; break;
        case SGZ_CH_SIDE: xr = (l - r) * w * 0.5f; break;
        case SGZ_CH_MIDSIDE: xr = (l + r) * w * 0.5f; xi = (l - r) * w * 0.5f; break;
        default: xr = l * w; xi = r * w; break;
        }
    }
    out[gid] = make_float2(xr, xi);
}

// Stockham autosort radix-2 DIF pass `s` (s = 0 .. log2N-1):  l = N >> (s+1), m = 1 << s
//   y[k + 2 j m]     =  x[k + j m] + x[k + j m + l m]
//   y[k + 2 j m + m] = (x[k + j m] - x[k + j m + l m]) * W_{2l}^j ,   W_{2l}^j = W_N^{j << s}
__global__ void __launch_bounds__(256)
genericStage(const float2 *x, float2 *y, const float2 *twN /*W_N^i, i < N/2*/, uint32_t N, uint32_t s, long ntasks)
{
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t half = N >> 1;
    if (gid >= size_t(ntasks) * half) return;
    const long t = long(gid / half);
    const uint32_t b = uint32_t(gid - size_t(t) * half);          // butterfly index = j * m + k
    const uint32_t m = 1u << s;
    const uint32_t k = b & (m - 1), j = b >> s;
    const float2 *xi = x + size_t(t) * N;
    float2 *yo = y + size_t(t) * N;
    const float2 c0 = xi[b], c1 = xi[b + half];                   // k + j m  and  k + j m + l m  (l m = N/2)
    const float2 w = twN[size_t(j) << s];
    const float dr = c0.x - c1.x, di = c0.y - c1.y;
    yo[k + 2 * j * m] = make_float2(c0.x + c1.x, c0.y + c1.y);
    yo[k + 2 * j * m + m] = make_float2(dr * w.x - di * w.y, dr * w.y + di * w.x);
}

__global__ void __launch_bounds__(256)
genericBins(const float2 *z, uint32_t N, uint32_t sides, uint32_t mode, long ntasks, float *bins /*[ntasks][N+1]*/)
{
    const size_t per = size_t(N) + 1;
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(ntasks) * per) return;
    const long t = long(gid / per);
    const uint32_t k = uint32_t(gid - size_t(t) * per);
    const float2 *Z = z + size_t(t) * N;
    float out;
    if (sides == 2) {                                             // Separate / MidSide, TransformDSP.inl:858-869
        if (k == 0) out = Z[0].x * 0.5f;
        else if (k == N) out = Z[0].y * 0.5f;
        else if (k == N / 2) out = 0.5f * __builtin_amdgcn_sqrtf(Z[k].x * Z[k].x + Z[k].y * Z[k].y);
        else {
            const float2 a = Z[k], b = Z[N - k];
            float u, v;
            if (k < N / 2) { u = a.x + b.x; v = a.y - b.y; }      // X1[k]
            else { u = a.x - b.x; v = a.y + b.y; }                // X2[N-k] (magnitude)
            out = 0.5f * __builtin_amdgcn_sqrtf(u * u + v * v);
            if (k == N / 2 - 1) out *= 0.5f;                      // quirk Q3, :864
        }
    } else {                                                      // mono :553-560 / complex :993-1002 (magnitudes)
        if (k == N) out = 0.f;
        else {
            out = __builtin_amdgcn_sqrtf(Z[k].x * Z[k].x + Z[k].y * Z[k].y);
            if (k == 0 || (k == N / 2 && mode != SGZ_CH_COMPLEX)) out *= 0.5f;
        }
    }
    bins[gid] = out;
}

// one thread per (task, side, pixel); exact fp32 order of the reference (contraction off)
__global__ void __launch_bounds__(256)
genericMap(const float *bins, uint32_t N, uint32_t P, uint32_t sides, const PixelRec *recs, const float *weights,
           float invSize, long ntasks, float *mapped /*[ntasks][sides][P]*/)
{
#pragma clang fp contract(off)
    const uint32_t total = sides * P;
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(ntasks) * total) return;
    const long t = long(gid / total);
    const uint32_t idx = uint32_t(gid - size_t(t) * total);
    const float *M = bins + size_t(t) * (size_t(N) + 1);
    const PixelRec rec = recs[idx];
    const int side = idx >= P ? 1 : 0;
    float val;
    if ((rec.kind & 1) == 0) {
        float acc = 0.f;
        int k = rec.a;
        for (int i = 0; i < rec.b; ++i) {
            const float prod = M[k] * weights[rec.c + i];
            acc = acc + prod;
            k = (k == int(N)) ? 0 : k + 1;
        }
        val = invSize * acc;
    } else {
        float best = 0.f;
        int arg = rec.c;
        for (int i = 0; i < rec.b; ++i) {
            const int off = rec.a + i;
            const int k = side ? (int(N) - off) : off;
            const float m = M[k];
            const float sq = m * m + 0.f;
            if (sq > best) { best = sq; arg = k; }
        }
        val = invSize * M[arg];
    }
    const float sq = val * val + 0.f;
    mapped[gid] = __builtin_sqrtf(sq);
}

// ---- SpectrumChannels::Phase (TransformDSP.inl:643-853): the bins stay complex ------------------------------------------
// csf after separateTransformsIPL and the DC / Nyquist fix-ups (:646-652): csf[k] = X1[k], csf[N-k] = X2[k] (1 <= k < N/2),
// csf[0] = Re Z[0] / 2, csf[N] = Im Z[0] / 2, csf[N/2] and csf[N/2-1] halved (quirk Q3).
__global__ void __launch_bounds__(256)
genericBinsPhase(const float2 *z, uint32_t N, long ntasks, float2 *csf /*[ntasks][N+1]*/)
{
#pragma clang fp contract(off)
    const size_t per = size_t(N) + 1;
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(ntasks) * per) return;
    const long t = long(gid / per);
    const uint32_t k = uint32_t(gid - size_t(t) * per);
    const float2 *Z = z + size_t(t) * N;
    float2 out;
    if (k == 0) out = make_float2(Z[0].x * 0.5f, 0.f);
    else if (k == N) out = make_float2(Z[0].y * 0.5f, 0.f);
    else if (k == N / 2) out = make_float2(0.5f * Z[k].x, 0.5f * Z[k].y);
    else {
        const uint32_t kk = k < N / 2 ? k : N - k;                  // the pair (kk, N - kk) is split together
        const float2 a = Z[kk], b = Z[N - kk];
        if (k < N / 2) out = make_float2((a.x + b.x) * 0.5f, (a.y - b.y) * 0.5f);     // X1[k]
        else out = make_float2((a.y + b.y) * 0.5f, (b.x - a.x) * 0.5f);               // X2[kk]
        if (k == N / 2 - 1) out = make_float2(0.5f * out.x, 0.5f * out.y);            // :652
    }
    csf[gid] = out;
}

// std::abs(std::complex<float>) = hypotf: glibc evaluates it as (float)sqrt((double)x*x + (double)y*y)
__device__ __forceinline__ float cabsHypot(float2 z)
{
    const double x = double(z.x), y = double(z.y);
    return float(sqrt(x * x + y * y));
}

// one thread per (task, pixel): wsp[2x] = magnitude -> plane 0, wsp[2x+1] = cancellation measure -> plane 1
__global__ void __launch_bounds__(256)
genericMapPhase(const float2 *csfAll, uint32_t N, uint32_t P, const PixelRec *recs, const float *weights, PhaseTables ph,
                float invSize, long ntasks, float *mapped /*[ntasks][2][P]*/)
{
#pragma clang fp contract(off)
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(ntasks) * P) return;
    const long t = long(gid / P);
    const uint32_t x = uint32_t(gid - size_t(t) * P);
    const float2 *csf = csfAll + size_t(t) * (size_t(N) + 1);
    const uint32_t type = ph.type[x];
    // value of csf[j] once bins below `norm` (and their mirrors) have been replaced by their magnitudes
    auto normalised = [&](int j, uint32_t norm) {
        const float2 v = csf[j];
        const bool isNorm = uint32_t(j) < norm || uint32_t(int(N) - j) < norm;
        return isNorm ? make_float2(cabsHypot(v), 0.f) : v;
    };
    auto filter = [&](const PixelRec &rec, uint32_t norm) {           // taps accumulate in order, per component
        float2 acc = make_float2(0.f, 0.f);
        int k = rec.a;
        for (int i = 0; i < rec.b; ++i) {
            const float2 v = normalised(k, norm);
            const float w = weights[rec.c + i];
            acc.x = acc.x + v.x * w;
            acc.y = acc.y + v.y * w;
            k = (k == int(N)) ? 0 : k + 1;
        }
        return acc;
    };
    float mag, cancel;
    if (type != 1u) {
        const PixelRec rl = recs[x], rr = recs[P + x];
        if (type == 0u) {
            const float2 iLeft = filter(rl, 0u), iRight = filter(rr, 0u);          // phase pass: un-normalised vectors
            const float2 sum = make_float2(iLeft.x + iRight.x, iLeft.y + iRight.y);
            const float cancellation = invSize * (ph.filtered ? __builtin_sqrtf(sum.x * sum.x + sum.y * sum.y) : cabsHypot(sum));
            const float mid = invSize * (cabsHypot(iLeft) + cabsHypot(iRight));
            cancel = 1.0f - (mid > 0.f ? (cancellation / mid) : 0.f);
            mag = mid;
        } else cancel = 0.f;                                          // never written by the reference (see oracle/spectrum.c)
        if (ph.filtered) {                                            // magnitude pass on the lazily normalised bins
            const uint32_t norm = ph.norm[x];
            const float2 iLeft = filter(rl, norm), iRight = filter(rr, norm);
            mag = invSize * (cabsHypot(iLeft) + cabsHypot(iRight));
        }
    } else {
        const PixelRec rec = recs[x];
        float maxValue = 0.f;
        int maxBin = rec.c;                                           // 0
        for (int i = 0; i < rec.b; ++i) {
            const int off = rec.a + i;
            const float2 l = normalised(off, ph.normFinal), r = normalised(int(N) - off, ph.normFinal);
            const float a = l.x * l.x + l.y * l.y, b = r.x * r.x + r.y * r.y;     // Math::square(complex) = |z|^2
            const float newMag = a < b ? b : a;                        // std::max
            if (newMag > maxValue) { maxValue = newMag; maxBin = off; }
        }
        const float2 leftMax = normalised(maxBin, ph.normFinal), rightMax = normalised(int(N) - maxBin, ph.normFinal);
        const float2 sum = make_float2(leftMax.x + rightMax.x, leftMax.y + rightMax.y);
        const float interference = invSize * cabsHypot(sum);
        const float mid = invSize * (cabsHypot(leftMax) + cabsHypot(rightMax));
        const float cancellation = interference / mid;
        mag = mid;
        cancel = 1.0f - (mid > 0.f ? cancellation : 0.f);
    }
    float *out = mapped + size_t(t) * 2 * P;
    out[x] = mag;
    out[P + x] = cancel;
}

static inline unsigned gridFor(size_t total) { return unsigned((total + 255) / 256); }

// Runs the generic path for tasks [0, ntasks) in slabs that fit the work buffers (work0/work1: complex [slab][N]).
hipError_t launchGeneric(const StftParams &prm, uint32_t N, const float2 *twN, float2 *work0, float2 *work1, float *binsWork,
                         long slab, hipStream_t stream, const PhaseTables *phase)
{
    const long tasks = prm.frames * long(prm.C);
    uint32_t log2N = 0;
    while ((1u << log2N) < N) ++log2N;
    for (long t0 = 0; t0 < tasks; t0 += slab) {
        const long nt = tasks - t0 < slab ? tasks - t0 : slab;
        if (phase) {
            // Phase: prepare -> FFT passes -> complex csf (in the first work buffer the passes leave free) -> map
            const float2 *csf;
            if (phase->csfIn == nullptr) {
                hipLaunchKernelGGL(genericPrepare, dim3(gridFor(size_t(nt) * N)), dim3(256), 0, stream, prm.planar, prm.chStride, prm.hop,
                                   prm.W, N, prm.C, prm.mode, prm.window, t0, nt, work0);
                float2 *src = work0, *dst = work1;
                for (uint32_t s = 0; s < log2N; ++s) {
                    hipLaunchKernelGGL(genericStage, dim3(gridFor(size_t(nt) * (N / 2))), dim3(256), 0, stream, src, dst, twN, N, s, nt);
                    float2 *tmp = src; src = dst; dst = tmp;
                }
                // complex csf needs N + 1 entries per task: the caller sized binsWork (float) as 2 * (N + 1) per task for Phase
                float2 *cout = phase->csfOut ? phase->csfOut + size_t(t0) * (size_t(N) + 1) : reinterpret_cast<float2 *>(binsWork);
                hipLaunchKernelGGL(genericBinsPhase, dim3(gridFor(size_t(nt) * (size_t(N) + 1))), dim3(256), 0, stream, src, N, nt, cout);
                csf = cout;
            } else {
                csf = phase->csfIn + size_t(t0) * (size_t(N) + 1);
            }
            if (prm.mapped)
                hipLaunchKernelGGL(genericMapPhase, dim3(gridFor(size_t(nt) * prm.P)), dim3(256), 0, stream, csf, N, prm.P, prm.recs,
                                   prm.weights, *phase, prm.invSize, nt, prm.mapped + size_t(t0) * 2 * prm.P);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            continue;
        }
        const float *bins;
        if (prm.binsIn == nullptr) {
            hipLaunchKernelGGL(genericPrepare, dim3(gridFor(size_t(nt) * N)), dim3(256), 0, stream, prm.planar, prm.chStride, prm.hop,
                               prm.W, N, prm.C, prm.mode, prm.window, t0, nt, work0);
            float2 *src = work0, *dst = work1;
            for (uint32_t s = 0; s < log2N; ++s) {
                hipLaunchKernelGGL(genericStage, dim3(gridFor(size_t(nt) * (N / 2))), dim3(256), 0, stream, src, dst, twN, N, s, nt);
                float2 *tmp = src; src = dst; dst = tmp;
            }
            float *bout = prm.binsOut ? prm.binsOut + size_t(t0) * (size_t(N) + 1) : binsWork;
            hipLaunchKernelGGL(genericBins, dim3(gridFor(size_t(nt) * (size_t(N) + 1))), dim3(256), 0, stream, src, N, prm.sides,
                               prm.mode, nt, bout);
            bins = bout;
        } else {
            bins = prm.binsIn + size_t(t0) * (size_t(N) + 1);
        }
        if (prm.mapped)
            hipLaunchKernelGGL(genericMap, dim3(gridFor(size_t(nt) * prm.sides * prm.P)), dim3(256), 0, stream, bins, N, prm.P, prm.sides,
                               prm.recs, prm.weights, prm.invSize, nt, prm.mapped + size_t(t0) * prm.sides * prm.P);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sgz
