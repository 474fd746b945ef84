// spectrum_generic.hip -- K_A for ANY power-of-two transform size (32 <= N, not only the fused 4096 / 32768
// kernel of spectrum_fft.hip): the same stages as separate HBM-resident kernels.  gfx950 only.
//
//   PrepSource     : prepareTransform (Source/Spectrum/TransformDSP.inl:39-231)   audio x window, read by the first FFT pass
//   genericStage   : one radix-16 (first pass: 2/4/8 if needed) Stockham autosort pass of the forward DFT, ping-pong in HBM
//                    (doTransform, :487-502 -- natural order, unnormalised)
//   genericBins    : two-for-one split + DC/Nyquist fix-ups + |.|  (:858-869 ; mono :553-560)  -> csf magnitudes [N+1]
//   genericMap     : pixel mapping from HBM-resident csf (:565-639, :871-985), same PixelRec table as the fused kernel
//
// This path moves ~(2 log16 N + 6) * N * 8 bytes per frame-pair through HBM/L2 instead of the fused kernel's
// 2*W*4: it exists for completeness (every window size the reference accepts works, e.g. BASELINE cfg5's
// N = 65536), not for speed; fusing N = 2 R^3 is listed under "next" in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "complex_dc.hpp"
#include "fft_common.hpp"
#include "kernels.hpp"

namespace sgz {

// prepareTransform (TransformDSP.inl:39-231): windowed, channel-mixed, zero-padded input sample n of task task0 + t
struct PrepSource {
    const float *planar;
    size_t chStride;
    uint32_t hop, W, C, mode;
    const float *window;
    long task0;
    // base of the left channel's frame for task task0 + t (the right channel is chStride further)
    __device__ __forceinline__ const float *frameOf(long t) const
    {
        const long task = task0 + t;
        const long frame = task / C;
        const uint32_t pair = uint32_t(task - frame * C);
        return planar + size_t(2 * pair) * chStride + size_t(frame) * hop;
    }
    __device__ __forceinline__ float2 sample(const float *L, uint32_t n) const
    {
        float xr = 0.f, xi = 0.f;
        if (n < W) {
            const float l = L[n], r = L[chStride + n], w = window[n];
            switch (mode) {                                       // TransformDSP.inl:59-216
            case SGZ_CH_LEFT: xr = l * w; break;
            case SGZ_CH_RIGHT: xr = r * w; break;
            case SGZ_CH_MERGE: xr = (l + r) * w * 0.5f; break;
            case SGZ_CH_SIDE: xr = (l - r) * w * 0.5f; break;
            case SGZ_CH_MIDSIDE: xr = (l + r) * w * 0.5f; xi = (l - r) * w * 0.5f; break;
            default: xr = l * w; xi = r * w; break;
            }
        }
        return make_float2(xr, xi);
    }
};

// Stockham autosort radix-RX DIF pass (RX = 2, 4, 8 or 16).  Before the pass the transform is N = RX * l * m with m the
// product of the earlier radices:   for j < l, k < m:
//   y[k + (RX j + q) m] = W_{RX l}^{j q} * sum_t x[k + j m + t l m] W_RX^{t q},   q < RX
// One thread per (j, k): RX strided loads (coalesced across threads), the RX-point DIF in registers (fft_common.hpp), one
// twiddle per output from the W_N table (W_{RX l}^{j q} = W_N^{j q m}; the table holds i < N/2, W_N^{i + N/2} = -W_N^i).
// Radix 16 moves the data through HBM/L2 log16 N times instead of log2 N.  FIRST: the pass reads the audio itself
// (window, channel mix and zero padding fused: no separate prepare pass).
template <int RX, bool FIRST>
__global__ void __launch_bounds__(256)
genericStage(const float2 *x, float2 *y, const float2 *twN /*W_N^i, i < N/2*/, uint32_t N, uint32_t m, long ntasks, const PrepSource prep)
{
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const uint32_t per = N / RX;                                  // butterflies per transform = l * m
    if (gid >= size_t(ntasks) * per) return;
    const long t = long(gid / per);
    const uint32_t b = uint32_t(gid - size_t(t) * per);           // j * m + k
    const uint32_t j = b / m, k = b - j * m;
    const float2 *xi = x + size_t(t) * N;
    float2 *yo = y + size_t(t) * N;
    float re[RX], im[RX];
    const float *frameL = FIRST ? prep.frameOf(t) : nullptr;
#pragma unroll
    for (int tt = 0; tt < RX; ++tt) {
        const float2 v = FIRST ? prep.sample(frameL, b + uint32_t(tt) * per) : xi[b + size_t(tt) * per];
        re[tt] = v.x; im[tt] = v.y;
    }
    dif<float, RX, RX, 0>(re, im);                                // output q at register brev(q)
    constexpr int LRX = RX == 16 ? 4 : (RX == 8 ? 3 : (RX == 4 ? 2 : 1));
#pragma unroll
    for (int q = 0; q < RX; ++q) {
        const int r = brev(q, LRX);
        float vr = re[r], vi = im[r];
        if (q > 0) {
            uint32_t idx = uint32_t((uint64_t(j) * uint32_t(q) * m) % N);     // exponent of W_N
            float sgn = 1.f;
            if (idx >= N / 2) { idx -= N / 2; sgn = -1.f; }
            const float2 w = twN[idx];
            const float wr = sgn * w.x, wi = sgn * w.y;
            const float tr = vr * wr - vi * wi;
            vi = vr * wi + vi * wr;
            vr = tr;
        }
        yo[k + (size_t(RX) * j + q) * m] = make_float2(vr, vi);
    }
}

__global__ void __launch_bounds__(256)
genericBins(const float2 *z, uint32_t N, uint32_t sides, uint32_t mode, long ntasks, float *bins /*[ntasks][N+1]*/,
            float2 *dcOut /*[ntasks][kSpecBins] or null: the csf entries that stay complex (complex_dc.hpp)*/)
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
    if (dcOut && k < N) {
        const int s = specSlot(mode, int(N), int(k));
        if (s >= 0) {
            const float f = specScale(mode, int(N), int(k));
            dcOut[size_t(t) * kSpecBins + s] = make_float2(f * Z[k].x, f * Z[k].y);
        }
    }
}

// 16 lanes per (task, side, pixel); exact fp32 order of the reference (contraction off).  An interpolated pixel is lane 0's
// sequential tap sum.  An arg-max pixel's bin run (1 .. several hundred bins at large N) is scanned 16 bins at a time with
// coalesced loads; every lane keeps the first strictly greater |X|^2 of its own subsequence and the lanes are merged with
// "larger square, then smaller scan offset" -- which is the reference's "first strictly greater in scan order".
constexpr int kMapLanes = 16;

__global__ void __launch_bounds__(256)
genericMap(const float *bins, uint32_t N, uint32_t P, uint32_t sides, const PixelRec *recs, const float *weights,
           float invSize, long ntasks, float *mapped /*[ntasks][sides][P]*/)
{
#pragma clang fp contract(off)
    const uint32_t total = sides * P;
    const size_t gid = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) / kMapLanes;
    const int lane = int(threadIdx.x) & (kMapLanes - 1);
    if (gid >= size_t(ntasks) * total) return;                   // whole 16-lane groups leave together
    const long t = long(gid / total);
    const uint32_t idx = uint32_t(gid - size_t(t) * total);
    const float *M = bins + size_t(t) * (size_t(N) + 1);
    const PixelRec rec = recs[idx];
    const int side = idx >= P ? 1 : 0;
    float val = 0.f;
    if ((rec.kind & 1) == 0) {
        // lane i fetches tap i and multiplies; the products are added in tap order (the reference's rounding)
        float prod = 0.f;
        if (lane < rec.b) {
            int k = rec.a + lane;
            k = k > int(N) ? k - (int(N) + 1) : k;                     // periodic over the N + 1 entries
            prod = M[k] * weights[rec.c + lane];
        }
        float acc = 0.f;
        for (int i = 0; i < rec.b; ++i) acc = acc + __shfl(prod, i, kMapLanes);
        val = invSize * acc;
    } else {
        float best = 0.f;
        int bestOff = 0x7fffffff;                                 // scan offset of this lane's winner (none yet)
        for (int i = lane; i < rec.b; i += kMapLanes) {
            const int off = rec.a + i;
            const int k = side ? (int(N) - off) : off;
            const float m = M[k];
            const float sq = m * m + 0.f;
            if (sq > best) { best = sq; bestOff = off; }
        }
#pragma unroll
        for (int d = 1; d < kMapLanes; d <<= 1) {
            const float ob = __shfl_xor(best, d, kMapLanes);
            const int oo = __shfl_xor(bestOff, d, kMapLanes);
            if (ob > best || (ob == best && oo < bestOff)) { best = ob; bestOff = oo; }
        }
        if (lane == 0) {
            const int arg = bestOff == 0x7fffffff ? rec.c : (side ? int(N) - bestOff : bestOff);   // nothing > 0: the run's `bin`
            val = invSize * M[arg];
        }
    }
    if (lane == 0) {
        const float sq = val * val + 0.f;
        mapped[gid] = __builtin_sqrtf(sq);
    }
}

// ---- SpectrumChannels::Phase (TransformDSP.inl:643-853): the bins stay complex ------------------------------------------
// one entry of Phase mode's csf from the raw transform (Zat(i) = Z[i]); see genericBinsPhase
template <typename ZAt>
__device__ __forceinline__ float2 phaseCsfEntry(uint32_t N, uint32_t k, ZAt Zat)
{
#pragma clang fp contract(off)
    float2 out;
    if (k == 0) { const float2 z0 = Zat(0); out = make_float2(z0.x * 0.5f, 0.f); }
    else if (k == N) { const float2 z0 = Zat(0); out = make_float2(z0.y * 0.5f, 0.f); }
    else if (k == N / 2) { const float2 zn = Zat(k); out = make_float2(0.5f * zn.x, 0.5f * zn.y); }
    else {
        const uint32_t kk = k < N / 2 ? k : N - k;                  // the pair (kk, N - kk) is split together
        const float2 a = Zat(kk), b = Zat(N - kk);
        if (k < N / 2) out = make_float2((a.x + b.x) * 0.5f, (a.y - b.y) * 0.5f);     // X1[k]
        else out = make_float2((a.y + b.y) * 0.5f, (b.x - a.x) * 0.5f);               // X2[kk]
        if (k == N / 2 - 1) out = make_float2(0.5f * out.x, 0.5f * out.y);            // :652
    }
    return out;
}

// csf after separateTransformsIPL and the DC / Nyquist fix-ups (:646-652): csf[k] = X1[k], csf[N-k] = X2[k] (1 <= k < N/2),
// csf[0] = Re Z[0] / 2, csf[N] = Im Z[0] / 2, csf[N/2] and csf[N/2-1] halved (quirk Q3).
__global__ void __launch_bounds__(256)
genericBinsPhase(const float2 *z, uint32_t N, long ntasks, float2 *csf /*[ntasks][N+1]*/, const uint32_t planar)
{
#pragma clang fp contract(off)
    const size_t per = size_t(N) + 1;
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(ntasks) * per) return;
    const long t = long(gid / per);
    const uint32_t k = uint32_t(gid - size_t(t) * per);
    // planar: the in-register FFT wrote the task's transform as two arrays, re[N] then im[N] (stftComplexKernel)
    const float2 *zt = z + size_t(t) * N;
    const float *zf = reinterpret_cast<const float *>(zt);
    auto Zat = [&](uint32_t i) { return planar ? make_float2(zf[i], zf[N + i]) : zt[i]; };
    const float2 out = phaseCsfEntry(N, k, Zat);
    csf[gid] = out;
}

// std::abs(std::complex<float>) = hypotf: glibc evaluates it as (float)sqrt((double)x*x + (double)y*y)
__device__ __forceinline__ float cabsHypot(float2 z)
{
    const double x = double(z.x), y = double(z.y);
    return float(sqrt(x * x + y * y));
}

// 16 lanes per (task, pixel) (see genericMap): wsp[2x] = magnitude -> plane 0, wsp[2x+1] = cancellation measure -> plane 1.
// FROMZ = 0: csfAll holds csf ([task][N+1], genericBinsPhase or the test hook); 1 / 2: csfAll holds the raw transforms
// ([task][N], interleaved / as re[N] im[N] planes) and every csf entry is split on the fly -- no csf array, no split pass.
template <int FROMZ>
__global__ void __launch_bounds__(256)
genericMapPhase(const float2 *csfAll, uint32_t N, uint32_t P, const PixelRec *recs, const float *weights, PhaseTables ph,
                float invSize, long ntasks, float *mapped /*[ntasks][2][P]*/)
{
#pragma clang fp contract(off)
    const size_t gid = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) / kMapLanes;
    const int lane = int(threadIdx.x) & (kMapLanes - 1);
    if (gid >= size_t(ntasks) * P) return;
    const long t = long(gid / P);
    const uint32_t x = uint32_t(gid - size_t(t) * P);
    const float2 *csf = csfAll + size_t(t) * (size_t(N) + (FROMZ ? 0 : 1));
    const float *zf = reinterpret_cast<const float *>(csf);
    auto entry = [&](int j) {
        if (FROMZ == 0) return csf[j];
        return phaseCsfEntry(N, uint32_t(j), [&](uint32_t i) { return FROMZ == 2 ? make_float2(zf[i], zf[N + i]) : csf[i]; });
    };
    const uint32_t type = ph.type[x];
    // value of csf[j] once bins below `norm` (and their mirrors) have been replaced by their magnitudes
    auto normalised = [&](int j, uint32_t norm) {
        const float2 v = entry(j);
        const bool isNorm = uint32_t(j) < norm || uint32_t(int(N) - j) < norm;
        return isNorm ? make_float2(cabsHypot(v), 0.f) : v;
    };
    // One filter window (<= 10 taps <= kMapLanes): lane i evaluates tap i -- the (possibly normalised) entry times its weight,
    // the expensive part -- and every lane then adds the products in tap order, so the sum rounds exactly like the sequential
    // loop of the reference.  All lanes of the pixel call it together.
    auto filter = [&](const PixelRec &rec, uint32_t norm) {
        float px = 0.f, py = 0.f;
        if (lane < rec.b) {
            int k = rec.a + lane;
            k = k > int(N) ? k - (int(N) + 1) : k;                     // periodic over the N + 1 entries
            const float2 v = normalised(k, norm);
            const float w = weights[rec.c + lane];
            px = v.x * w;
            py = v.y * w;
        }
        float2 acc = make_float2(0.f, 0.f);
        for (int i = 0; i < rec.b; ++i) {
            acc.x = acc.x + __shfl(px, i, kMapLanes);
            acc.y = acc.y + __shfl(py, i, kMapLanes);
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
        if (lane != 0) return;
    } else {
        const PixelRec rec = recs[x];
        float maxValue = 0.f;
        int maxBin = 0x7fffffff;
        for (int i = lane; i < rec.b; i += kMapLanes) {
            const int off = rec.a + i;
            float2 l, r;
            if (FROMZ != 0 && off > 0 && off < int(N / 2) - 1) {
                // csf[off] = X1[off] and csf[N - off] = X2[off] come from the same two transform values: load them once (the general
                // path below fetches Z[off] and Z[N - off] for either entry), same expressions as phaseCsfEntry
                const float2 za = FROMZ == 2 ? make_float2(zf[off], zf[N + off]) : csf[off];
                const float2 zb = FROMZ == 2 ? make_float2(zf[N - off], zf[N + (N - off)]) : csf[N - off];
                l = make_float2((za.x + zb.x) * 0.5f, (za.y - zb.y) * 0.5f);
                r = make_float2((za.y + zb.y) * 0.5f, (zb.x - za.x) * 0.5f);
                if (uint32_t(off) < ph.normFinal) {                     // (off < N - off: the one condition covers both entries)
                    l = make_float2(cabsHypot(l), 0.f);
                    r = make_float2(cabsHypot(r), 0.f);
                }
            } else {
                l = normalised(off, ph.normFinal);
                r = normalised(int(N) - off, ph.normFinal);
            }
            const float a = l.x * l.x + l.y * l.y, b = r.x * r.x + r.y * r.y;     // Math::square(complex) = |z|^2
            const float newMag = a < b ? b : a;                        // std::max
            if (newMag > maxValue) { maxValue = newMag; maxBin = off; }
        }
#pragma unroll
        for (int d = 1; d < kMapLanes; d <<= 1) {                      // larger value, then smaller offset = first strictly greater
            const float ov = __shfl_xor(maxValue, d, kMapLanes);
            const int ob = __shfl_xor(maxBin, d, kMapLanes);
            if (ov > maxValue || (ov == maxValue && ob < maxBin)) { maxValue = ov; maxBin = ob; }
        }
        if (lane != 0) return;
        if (maxBin == 0x7fffffff) maxBin = rec.c;                      // 0 (:813)
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

// all passes of the forward FFT, ping-ponging between the two work buffers; on return `src` holds the spectrum.
// Radix 16 passes, preceded by one pass of radix 2, 4 or 8 when log2 N is not a multiple of 4.
// The first pass reads the audio (prep) and writes `dst`; the passes ping-pong from there; on return `src` holds the transform.
static void runStages(const PrepSource &prep, float2 *&src, float2 *&dst, const float2 *twN, uint32_t N, uint32_t log2N, long nt,
                      hipStream_t stream)
{
    uint32_t m = 1, left = log2N;
    bool first = true;
    auto launch = [&](auto rx, auto isFirst, size_t threads) {
        hipLaunchKernelGGL((genericStage<decltype(rx)::value, decltype(isFirst)::value>), dim3(gridFor(threads)), dim3(256), 0, stream, src,
                           dst, twN, N, m, nt, prep);
    };
    auto pass = [&](int lr) {
        const size_t threads = size_t(nt) * (N >> lr);
        using T = std::true_type; using F = std::false_type;
        switch (lr) {
        case 1: first ? launch(std::integral_constant<int, 2>{}, T{}, threads) : launch(std::integral_constant<int, 2>{}, F{}, threads); break;
        case 2: first ? launch(std::integral_constant<int, 4>{}, T{}, threads) : launch(std::integral_constant<int, 4>{}, F{}, threads); break;
        case 3: first ? launch(std::integral_constant<int, 8>{}, T{}, threads) : launch(std::integral_constant<int, 8>{}, F{}, threads); break;
        default: first ? launch(std::integral_constant<int, 16>{}, T{}, threads) : launch(std::integral_constant<int, 16>{}, F{}, threads); break;
        }
        first = false;
        m <<= lr; left -= uint32_t(lr);
        float2 *tmp = src; src = dst; dst = tmp;
    };
    if (left % 4) pass(int(left % 4));
    while (left) pass(4);
}

// the pixels that reach a csf entry the reference leaves complex, redone with those entries (complex_dc.hpp); split: csf in
// the halves path's even / odd layout
__global__ void __launch_bounds__(64)
complexDcFixKernel(const float *bins, const float2 *dc, uint32_t N, uint32_t P, uint32_t mode, uint32_t split, const PixelRec *recs,
                   const float *weights, const uint32_t *dcPixels, uint32_t nDc, float invSize, long ntasks,
                   float *mapped /*[ntasks][P]*/)
{
    const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (gid >= size_t(ntasks) * nDc) return;
    const long t = long(gid / nDc);
    const uint32_t x = dcPixels[gid - size_t(t) * nDc];
    const float *M = bins + size_t(t) * (size_t(N) + 1);
    const float2 *spec = dc + size_t(t) * kSpecBins;
    const int half = int(N >> 1);
    mapped[size_t(t) * P + x] = complexDcPixel(
        recs[x], weights, invSize, int(N), mode,
        [&](int k) { return M[!split ? k : ((k & 1) ? half + 1 + (k >> 1) : (k >> 1))]; }, [&](int s) { return spec[s]; });
}

hipError_t launchComplexDcFix(const StftParams &prm, uint32_t N, const float *bins, const float2 *dc, long ntasks, float *mapped,
                              hipStream_t stream)
{
    if (prm.nDcPixels == 0 || dc == nullptr) return hipSuccess;       // (the listed pixels are all on side 0: sides == 1 in these modes)
    const size_t n = size_t(ntasks) * prm.nDcPixels;
    hipLaunchKernelGGL(complexDcFixKernel, dim3(unsigned((n + 63) / 64)), dim3(64), 0, stream, bins, dc, N, prm.P, prm.mode, prm.binsSplit,
                       prm.recs, prm.weights, prm.dcPixels, prm.nDcPixels, prm.invSize, ntasks, mapped);
    return hipGetLastError();
}

hipError_t launchGenericMap(const StftParams &prm, uint32_t N, const float *bins, long ntasks, float *mapped, hipStream_t stream)
{
    hipLaunchKernelGGL(genericMap, dim3(gridFor(size_t(ntasks) * prm.sides * prm.P * kMapLanes)), dim3(256), 0, stream, bins, N, prm.P,
                       prm.sides, prm.recs, prm.weights, prm.invSize, ntasks, mapped);
    return hipGetLastError();
}

// Runs the generic path for tasks [0, ntasks) in slabs that fit the work buffers (work0/work1: complex [slab][N]).
hipError_t launchGeneric(const StftParams &prm, uint32_t N, const float2 *twN, float2 *work0, float2 *work1, float *binsWork,
                         long slab, hipStream_t stream, const PhaseTables *phase, bool sideMap)
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
                const PrepSource prep{prm.planar, prm.chStride, prm.hop, prm.W, prm.C, prm.mode, prm.window, t0};
                float2 *src = work0, *dst = work1;
                if (phase->fusedFft) {                                  // N = R^3: the in-register FFT writes Z
                    StftParams pz = prm;
                    pz.taskBase = t0;
                    pz.frames = nt / long(prm.C);
                    pz.zOut = work0;
                    hipError_t e2 = launchStftComplex(pz, N, int(nt), stream);
                    if (e2 != hipSuccess) return e2;
                } else
                    runStages(prep, src, dst, twN, N, log2N, nt, stream);
                if (phase->csfOut == nullptr) {
                    // production: the map kernel splits the csf entries it needs out of Z (no csf array, no split pass)
                    if (prm.mapped) {
                        const dim3 grid(gridFor(size_t(nt) * prm.P * kMapLanes));
                        float *out = prm.mapped + size_t(t0) * 2 * prm.P;
                        if (phase->fusedFft)
                            hipLaunchKernelGGL(genericMapPhase<2>, grid, dim3(256), 0, stream, src, N, prm.P, prm.recs, prm.weights, *phase,
                                               prm.invSize, nt, out);
                        else
                            hipLaunchKernelGGL(genericMapPhase<1>, grid, dim3(256), 0, stream, src, N, prm.P, prm.recs, prm.weights, *phase,
                                               prm.invSize, nt, out);
                    }
                    hipError_t e = hipGetLastError();
                    if (e != hipSuccess) return e;
                    continue;
                }
                // test hook (complex bins out): csf needs N + 1 entries per task
                float2 *cout = phase->csfOut + size_t(t0) * (size_t(N) + 1);
                hipLaunchKernelGGL(genericBinsPhase, dim3(gridFor(size_t(nt) * (size_t(N) + 1))), dim3(256), 0, stream, src, N, nt, cout,
                                   phase->fusedFft);
                csf = cout;
            } else {
                csf = phase->csfIn + size_t(t0) * (size_t(N) + 1);
            }
            if (prm.mapped)
                hipLaunchKernelGGL(genericMapPhase<0>, dim3(gridFor(size_t(nt) * prm.P * kMapLanes)), dim3(256), 0, stream, csf, N, prm.P,
                                   prm.recs, prm.weights, *phase, prm.invSize, nt, prm.mapped + size_t(t0) * 2 * prm.P);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            continue;
        }
        const float *bins;
        if (prm.binsIn == nullptr) {
            const PrepSource prep{prm.planar, prm.chStride, prm.hop, prm.W, prm.C, prm.mode, prm.window, t0};
            float2 *src = work0, *dst = work1;
            runStages(prep, src, dst, twN, N, log2N, nt, stream);
            float *bout = prm.binsOut ? prm.binsOut + size_t(t0) * (size_t(N) + 1) : binsWork;
            hipLaunchKernelGGL(genericBins, dim3(gridFor(size_t(nt) * (size_t(N) + 1))), dim3(256), 0, stream, src, N, prm.sides,
                               prm.mode, nt, bout, prm.dcOut);
            bins = bout;
        } else {
            bins = prm.binsIn + size_t(t0) * (size_t(N) + 1);
        }
        if (prm.mapped && sideMap && mapSidesFit(prm, N)) {
            // LDS-staged per-side map (spectrum_fft.hip): one coalesced pass over csf instead of 16-lane gathers per pixel
            StftParams p2 = prm;
            p2.binsSplit = 0;
            hipError_t e2 = launchMapSides(p2, N, bins, nt, prm.mapped + size_t(t0) * prm.sides * prm.P, stream);
            if (e2 == hipSuccess && prm.binsIn == nullptr)
                e2 = launchComplexDcFix(p2, N, bins, prm.dcOut, nt, prm.mapped + size_t(t0) * prm.sides * prm.P, stream);
            if (e2 != hipSuccess) return e2;
        } else if (prm.mapped) {
            hipLaunchKernelGGL(genericMap, dim3(gridFor(size_t(nt) * prm.sides * prm.P * kMapLanes)), dim3(256), 0, stream, bins, N, prm.P, prm.sides,
                               prm.recs, prm.weights, prm.invSize, nt, prm.mapped + size_t(t0) * prm.sides * prm.P);
            if (prm.binsIn == nullptr) {
                hipError_t e2 = launchComplexDcFix(prm, N, bins, prm.dcOut, nt, prm.mapped + size_t(t0) * prm.sides * prm.P, stream);
                if (e2 != hipSuccess) return e2;
            }
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace sgz
