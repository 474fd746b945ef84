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
// contraction is off in these functions (NB: hip's __fmul_rn/__fadd_rn are plain * and + and would be fused,
// and __fsqrt_rn is the approximate native sqrt -- neither is used here).
//
// The reference's arg-max scan over a bin run ("first strictly greater |X|^2 wins", :957-979) is sequential and the
// runs are very uneven (1 .. ~140 bins at the top of a log view).  Here every run is cut into pieces of <= 16 bins
// (plan.cpp, MaxItem); a thread scans one piece with 16 independent LDS reads and merges through ds_max_u64 on the
// key (bits(|X|^2) << 32 | ~offset): larger square wins, equal squares -> smaller scan offset wins, which is
// exactly "first strictly greater".  Key 0 (no square > 0) falls back to `bin` like the reference's initial value.
template <int LR>
__device__ __forceinline__ float finishPixel(float val)
{
#pragma clang fp contract(off)
    // mapAndTransformDFTFilters: magnitude = sqrt(re*re + im*im), im == 0 (TransformDSP.inl:1331,:1365)
    const float sq = val * val + 0.f;
    return __builtin_sqrtf(sq);                                        // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
}

template <int LR>
__device__ __forceinline__ void mapPixelsSerial(const StftParams &prm, const float *lds, int tid, long task)
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
        out[idx] = finishPixel<LR>(val);
    }
}

// balanced version; `slots` = sides*P zero-initialised 64-bit keys in LDS.  Contains one workgroup barrier.
template <int LR>
__device__ __forceinline__ void mapPixelsBalanced(const StftParams &prm, const float *lds, unsigned long long *slots,
                                                  int tid, long task)
{
#pragma clang fp contract(off)
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const int total = int(prm.sides * prm.P);
    float *out = prm.mapped + size_t(task) * total;
    // (a) arg-max pieces
    for (uint32_t it = tid; it < prm.nItems; it += T) {
        const MaxItem item = prm.items[it];
        const int off0 = int(item.off0cnt & 0xFFFFFFu), cnt = int(item.off0cnt >> 24);
        const bool right = item.slot >= prm.P;
        float best = 0.f;
        uint32_t bestOff = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i < cnt) {
                const int off = off0 + i;
                const int k = right ? (N - off) : off;
                const float m = lds[k + (k >> LR)];
                const float sq = m * m + 0.f;                         // Math::square(csf[offset]) with imag == 0
                if (sq > best) { best = sq; bestOff = uint32_t(off); }
            }
        }
        if (best > 0.f) {
            const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(best)) << 32) | (0xFFFFFFFFu - bestOff);
            atomicMax(&slots[item.slot], key);
        }
    }
    // (b) interpolated pixels (<= 10 taps, accumulated in tap order)
    for (int idx = tid; idx < total; idx += T) {
        const PixelRec rec = prm.recs[idx];
        if (rec.kind != 0) continue;
        float acc = 0.f;
        int k = rec.a;
#pragma unroll
        for (int i = 0; i < kMaxTaps; ++i) {
            if (i < rec.b) {
                const float m = lds[k + (k >> LR)];
                const float prod = m * prm.weights[rec.c + i];
                acc = acc + prod;
                k = (k == N) ? 0 : k + 1;
            }
        }
        out[idx] = finishPixel<LR>(prm.invSize * acc);
    }
    __syncthreads();
    // (c) resolve the arg-max pixels
    for (int idx = tid; idx < total; idx += T) {
        const PixelRec rec = prm.recs[idx];
        if (rec.kind != 1) continue;
        const unsigned long long key = slots[idx];
        int k = rec.c;                                                   // maxLBin = maxRBin = bin (TransformDSP.inl:953)
        if ((key >> 32) != 0) {
            const int off = int(0xFFFFFFFFu - uint32_t(key));
            k = (idx >= int(prm.P)) ? (N - off) : off;
        }
        out[idx] = finishPixel<LR>(prm.invSize * lds[k + (k >> LR)]);
    }
}

// An opaque copy of a per-thread constant: address arithmetic derived from it cannot be hoisted out of the frame
// loop (where it would pin VGPRs for the whole iteration and spill); it is recomputed where it is used instead.
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

#define SGZ_CLK(slot)                                                                                   \
    do {                                                                                                \
        if (prm.phaseClock && tid == 0 && blockIdx.x == 0 && task == 0)                   \
            prm.phaseClock[slot] = __builtin_readcyclecounter();                                         \
    } while (0)

// Factorised twiddles: W^{x q} for q = 4a + b is B_a * A_b with A_b = W^{x b} (b = 1..3) and B_a = W^{x 4a}
// (a = 1..R/4-1), so a thread fetches 3 + R/4 - 1 complex values instead of R - 1 (10 instead of 31 at R = 32)
// and spends 4 VALU ops per product.  tw: table rows [A_1, A_2, A_3, B_1, .., B_{R/4-1}], row stride `rowBytes`.
template <int LR>
struct TwFactors {
    static constexpr int R = 1 << LR;
    static constexpr int NB = R / 4 - 1;
    float2 a[3];
    float2 b[NB];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int voff, int rowBytes)
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = bufLoad2(rs, voff, i * rowBytes);
#pragma unroll
        for (int i = 0; i < NB; ++i) b[i] = bufLoad2(rs, voff, (3 + i) * rowBytes);
    }
    // multiply the DIF output (bit-reversed order) by W^{x q}, q = 1..R-1
    __device__ __forceinline__ void apply(float (&re)[R], float (&im)[R]) const
    {
#pragma unroll
        for (int q = 1; q < R; ++q) {
            const int qa = q >> 2, qb = q & 3;
            float wx, wy;
            if (qa == 0) { wx = a[qb - 1].x; wy = a[qb - 1].y; }
            else if (qb == 0) { wx = b[qa - 1].x; wy = b[qa - 1].y; }
            else {
                wx = b[qa - 1].x * a[qb - 1].x - b[qa - 1].y * a[qb - 1].y;
                wy = b[qa - 1].x * a[qb - 1].y + b[qa - 1].y * a[qb - 1].x;
            }
            const int i = brev(q, LR);
            const float x = re[i], y = im[i];
            re[i] = x * wx - y * wy;
            im[i] = x * wy + y * wx;
        }
    }
};

// One workgroup = one (frame, pair) at a time.  LR = log2(R).
//
// Thread roles.  pass 1: t = tid.  passes 2/3: group gi = tid / R owns q = qOf(gi), lane l = tid % R is t2 (pass 2)
// then q2 (pass 3).  q's are paired {p, R-p} (and {0, R/2}) on sibling groups of ONE wave, so the k <-> N-k mirror
// of the two-for-one split is a lane permutation inside the wave (ds_bpermute), not a workgroup exchange.
template <int LR>
__global__ void __launch_bounds__(1 << (2 * LR))
stftMapKernel(const StftParams prm)
{
    constexpr int R = 1 << LR;
    constexpr int T = R * R;
    constexpr int N = R * T;
    constexpr int PADSTRIDE = T + (T >> LR);          // padded distance between k and k + T
    constexpr int SCRATCH = N + (N >> LR) + 4;        // float index of thread 0's 2R-float scratch
    constexpr int SLOTS = SCRATCH + 2 * R + 4;        // float index (even) of the arg-max slots (sides*P u64)
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const long tasks = prm.frames * long(prm.C);
    const int voff4 = tid * 4, voff8 = tid * 8;
    const int gi = tid >> LR, l = tid & (R - 1);
    const int pairIdx = gi >> 1;
    const int q = (gi & 1) ? (pairIdx == 0 ? R / 2 : R - pairIdx) : pairIdx;
    // mirror partner thread (same wave): (q', q2') with q' = (R - q) % R
    const int pg = gi < 2 ? gi : (gi ^ 1);
    const int pl = gi == 0 ? ((R - l) & (R - 1)) : (R - 1 - l);
    const int partnerLaneBytes = (((pg << LR) + pl) & 63) << 2;
    const bool split = (prm.sides == 2);
    const int mode = prm.mode;
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(lds + SLOTS);
    const bool balanced = prm.items != nullptr;
    // prepareTransform channel mixes (TransformDSP.inl:59-216): re = (mixRL*L + mixRR*R)*w*mixS, im likewise
    float mixRL = 1.f, mixRR = 0.f, mixIL = 0.f, mixIR = 1.f, mixS = 1.f;      // Phase / Separate / Complex
    if (mode == SGZ_CH_LEFT) { mixIR = 0.f; }
    else if (mode == SGZ_CH_RIGHT) { mixRL = 0.f; mixRR = 1.f; mixIR = 0.f; }
    else if (mode == SGZ_CH_MERGE) { mixRR = 1.f; mixIR = 0.f; mixS = 0.5f; }
    else if (mode == SGZ_CH_SIDE) { mixRR = -1.f; mixIR = 0.f; mixS = 0.5f; }
    else if (mode == SGZ_CH_MIDSIDE) { mixRR = 1.f; mixIL = 1.f; mixIR = -1.f; mixS = 0.5f; }

    float re[R], im[R];
    // audio of a task goes in flight early (before the previous frame's mapping phase), the window after it:
    // at most 2R + O(20) registers are live across the mapping loop, 3R only once its registers are dead.
    auto issueAudio = [&](long task) {
        const long frame = task / prm.C;
        const int pair = int(task - frame * prm.C);
        const float *L = prm.planar + size_t(2 * pair) * prm.chStride + size_t(frame) * prm.hop;
        const __amdgpu_buffer_rsrc_t rsL = makeRsrc(L, prm.W * 4u);
        const __amdgpu_buffer_rsrc_t rsR = makeRsrc(L + prm.chStride, prm.W * 4u);
        if (prm.ablate & 64) {     // EXPERIMENT (timing only, wrong layout): same bytes through 16 B/lane loads
#pragma unroll
            for (int j = 0; j < R; j += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(L + (j / 4) * (T * 4) + tid * 4);
                const float4 b = *reinterpret_cast<const float4 *>(L + prm.chStride + (j / 4) * (T * 4) + tid * 4);
                re[j] = a.x; re[j + 1] = a.y; re[j + 2] = a.z; re[j + 3] = a.w;
                im[j] = b.x; im[j + 1] = b.y; im[j + 2] = b.z; im[j + 3] = b.w;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            re[j] = bufLoad(rsL, voff4, j * (T * 4));
            im[j] = bufLoad(rsR, voff4, j * (T * 4));
        }
    };
    // window the prefetched samples (prepareTransform, TransformDSP.inl:59-216); the window itself is L2 resident
    auto applyWindow = [&]() {
        const __amdgpu_buffer_rsrc_t rsW = makeRsrc(prm.window, prm.W * 4u);
#pragma unroll
        for (int jb = 0; jb < R; jb += 16) {
            float wv[16];
            if (prm.ablate & 64) {
#pragma unroll
                for (int j = 0; j < 16; j += 4) {
                    const float4 a = *reinterpret_cast<const float4 *>(prm.window + ((jb + j) / 4) * (T * 4) + tid * 4);
                    wv[j] = a.x; wv[j + 1] = a.y; wv[j + 2] = a.z; wv[j + 3] = a.w;
                }
            } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) wv[j] = bufLoad(rsW, voff4, (jb + j) * (T * 4));
            }
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = jb + jj;
                const float lft = re[j], rgt = im[j], w = (prm.ablate & 128) ? 1.0f : wv[jj];
                // branch-free channel mix: (a*l + b*r) * w * s with a, b in {0, +-1}, s in {1, 0.5} rounds exactly
                // like the reference's `(l +- r) * w * 0.5f` / `l * w` (adding a signed zero is exact)
                const float xr = (mixRL * lft + mixRR * rgt) * w * mixS;
                const float xi = (mixIL * lft + mixIR * rgt) * w * mixS;
                re[j] = xr; im[j] = xi;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // XCD-aware task order.  Workgroup b is observed to run on XCD b % 8 (a speed assumption only, never a
    // correctness one): give every XCD one contiguous eighth of the frames, so that the 32 workgroups sharing an
    // L2 walk 32 consecutive (75 %-overlapping) frames together and each sample is fetched from HBM once per XCD.
    const bool xcdOrder = (gridDim.x % 8 == 0) && tasks >= 16;
    const long perXcd = xcdOrder ? (tasks + 7) / 8 : tasks;
    const long stride = xcdOrder ? gridDim.x / 8 : gridDim.x;
    const long xcdBase = xcdOrder ? long(blockIdx.x % 8) * perXcd : 0;
    long local = xcdOrder ? long(blockIdx.x / 8) : long(blockIdx.x);
    auto taskAt = [&](long loc) { return (loc < perXcd && xcdBase + loc < tasks) ? xcdBase + loc : -1L; };
    long task = taskAt(local);
    if (task >= 0 && prm.binsIn == nullptr) { issueAudio(task); applyWindow(); }

    for (; task >= 0; local += stride, task = taskAt(local)) {
        const long nextTask = taskAt(local + stride);
        SGZ_CLK(0);
        if (prm.binsIn == nullptr) {
            // ------------------------------------------------------------------ pass 1: DIF (samples already windowed)
            SGZ_CLK(1);
            {
                if (!(prm.ablate & 1)) dif<R, R, 0>(re, im);
                __builtin_amdgcn_sched_barrier(0);
                if (!(prm.ablate & 32)) {
                TwFactors<LR> tw;
                tw.load(makeRsrc(prm.tw1, uint32_t(3 + R / 4 - 1) * T * 8u), voff8, T * 8);
                tw.apply(re, im);                                      // times W_N^{t q}
                }
            }
            SGZ_CLK(2);
            // ---------------------------------------------------------- exchange 1 (workgroup-wide; re then im)
            __syncthreads();                                           // previous frame's mapping reads are done
            if (!(prm.ablate & 2)) {
            const int tw_ = opaque(tid);                               // write index
            const int rd_ = opaque(q * T + l);                         // read base
#pragma unroll
            for (int qq = 0; qq < R; ++qq) lds[qq * T + tw_] = re[brev(qq, LR)];
            __syncthreads();
#pragma unroll
            for (int j2 = 0; j2 < R; ++j2) re[j2] = lds[rd_ + R * j2];
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < R; ++qq) lds[qq * T + tw_] = im[brev(qq, LR)];
            __syncthreads();
#pragma unroll
            for (int j2 = 0; j2 < R; ++j2) im[j2] = lds[rd_ + R * j2];
            }
            SGZ_CLK(3);
            // ------------------------------------------------------------------ pass 2 (thread (q, t2 = l))
            {
                if (!(prm.ablate & 1)) dif<R, R, 0>(re, im);
                __builtin_amdgcn_sched_barrier(0);
                if (!(prm.ablate & 32)) {
                TwFactors<LR> tw;
                tw.load(makeRsrc(prm.tw2, uint32_t(3 + R / 4 - 1) * R * 8u), l * 8, R * 8);
                tw.apply(re, im);                                      // times W_T^{t2 q2}
                }
            }
            SGZ_CLK(4);
            // ------------------------------- exchange 2: R x R transpose inside each R-lane group (wave-local)
            const int l2_ = opaque(l);
            const int base = opaque(gi) * (R * (R + 1));
            __syncthreads();                                           // every wave has finished reading exchange 1
            if (!(prm.ablate & 4)) {
#pragma unroll
            for (int q2 = 0; q2 < R; ++q2) lds[base + q2 * (R + 1) + l2_] = re[brev(q2, LR)];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int j = 0; j < R; ++j) re[j] = lds[base + l2_ * (R + 1) + j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q2 = 0; q2 < R; ++q2) lds[base + q2 * (R + 1) + l2_] = im[brev(q2, LR)];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int j = 0; j < R; ++j) im[j] = lds[base + l2_ * (R + 1) + j];
            }
            SGZ_CLK(5);
            // ------------------------------------------------------------------ pass 3 (thread (q, q2 = l))
            if (!(prm.ablate & 1)) dif<R, R, 0>(re, im);
            SGZ_CLK(6);
            // X[c + T m3] at index brev(m3), c = q + R q2
            const int c = opaque(q + R * l);
            const int ownBase = c + (c >> LR);                         // padded address of k = c
            const int tid3 = opaque(tid);
            const int plb = opaque(partnerLaneBytes);
            if (split && !(prm.ablate & 8)) {
                if (tid == 0) {                                        // column c = 0 mirrors onto itself: redo it below
#pragma unroll
                    for (int m3 = 0; m3 < R; ++m3) {
                        lds[SCRATCH + 2 * m3] = re[brev(m3, LR)];
                        lds[SCRATCH + 2 * m3 + 1] = im[brev(m3, LR)];
                    }
                }
                // k = c + T m3 pairs with N - k = (T - c) + T (R-1-m3): partner thread, register R-1-m3
#pragma unroll
                for (int m3 = 0; m3 < R / 2; ++m3) {
                    const int ia = brev(m3, LR), ib = brev(R - 1 - m3, LR);
                    const float mra = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plb, __builtin_bit_cast(int, re[ib])));
                    const float mrb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plb, __builtin_bit_cast(int, re[ia])));
                    const float mia = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plb, __builtin_bit_cast(int, im[ib])));
                    const float mib = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plb, __builtin_bit_cast(int, im[ia])));
                    // k < N/2: X1 = (Z[k] + conj Z[N-k])/2 ; k > N/2: X2 = (Z[N-k'] - conj Z[k'])/(2i) (see oracle)
                    const float ua = re[ia] + mra, va = im[ia] - mia;
                    const float ub = re[ib] - mrb, vb = im[ib] + mib;
                    re[ia] = 0.5f * __builtin_amdgcn_sqrtf(ua * ua + va * va);
                    re[ib] = 0.5f * __builtin_amdgcn_sqrtf(ub * ub + vb * vb);
                }
            } else {
                if (tid == 0) { lds[SCRATCH] = re[0]; lds[SCRATCH + 1] = im[0];
                                lds[SCRATCH + R] = re[brev(R / 2, LR)]; lds[SCRATCH + R + 1] = im[brev(R / 2, LR)]; }
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) {                       // csf[k] = |Z[k]| (TransformDSP.inl:553-560, :993-1002)
                    const int i = brev(m3, LR);
                    re[i] = __builtin_amdgcn_sqrtf(re[i] * re[i] + im[i] * im[i]);
                }
            }
            SGZ_CLK(7);
            __syncthreads();                                           // exchange-2 tiles are dead: M may overwrite them
#pragma unroll
            for (int m3 = 0; m3 < R; ++m3) lds[ownBase + m3 * PADSTRIDE] = re[brev(m3, LR)];
            __syncthreads();
            if (balanced)
                for (int i = tid; i < int(prm.sides * prm.P); i += T) slots[i] = 0ull;
            if (split && tid >= 1 && tid < R / 2) {
                // column 0: k = T m3 pairs with T (R - m3); both were held by thread 0 -> lanes 1..R/2-1 redo them
                const int m3 = tid;
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
                    lds[N / 2 + ((N / 2) >> LR)] = 0.5f * __builtin_amdgcn_sqrtf(nyRe * nyRe + nyIm * nyIm);   // :863
                } else {
                    lds[N + (N >> LR)] = 0.f;
                    lds[0] = 0.5f * __builtin_amdgcn_sqrtf(dcRe * dcRe + dcIm * dcIm);
                    if (mode != SGZ_CH_COMPLEX)
                        lds[N / 2 + ((N / 2) >> LR)] = 0.5f * __builtin_amdgcn_sqrtf(nyRe * nyRe + nyIm * nyIm);
                }
            }
            if (split && tid == T - 1) {
                const int kq = N / 2 - 1;                            // held by thread (q2 = R-1, q = R-1)... any thread may scale it
                lds[kq + (kq >> LR)] *= 0.5f;                        // csf[N/2-1] *= 0.5 (quirk Q3, :864)
            }
            __syncthreads();
        } else {
            // test path (sgz_stage_map_from_bins): csf magnitudes come from HBM
            const float *src = prm.binsIn + size_t(task) * (N + 1);
            __syncthreads();
            for (int k = tid; k <= N; k += T) lds[k + (k >> LR)] = src[k];
            if (balanced)
                for (int i = tid; i < int(prm.sides * prm.P); i += T) slots[i] = 0ull;
            __syncthreads();
        }
        SGZ_CLK(8);

        if (prm.binsOut) {
            float *dst = prm.binsOut + size_t(task) * (N + 1);
            for (int k = tid; k <= N; k += T) dst[k] = lds[k + (k >> LR)];
        }
        // next frame's audio goes in flight now and lands during the mapping phase
        const bool more = nextTask >= 0 && prm.binsIn == nullptr;
        if (more) issueAudio(nextTask);
        SGZ_CLK(9);
        // ---------------------------------------------------------------------- pixel mapping
        if (prm.mapped && !(prm.ablate & 16)) {
            if (balanced) mapPixelsBalanced<LR>(prm, lds, slots, tid, task);
            else mapPixelsSerial<LR>(prm, lds, tid, task);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) applyWindow();
        SGZ_CLK(10);
    }
}

template <int LR>
static hipError_t launchStft(const StftParams &prm, int grid, hipStream_t stream)
{
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const size_t baseBytes = (size_t(N) + (N >> LR) + 4 + 2 * R + 4) * sizeof(float);
    const size_t slotBytes = size_t(prm.sides) * prm.P * 8;
    StftParams p2 = prm;
    size_t ldsBytes = baseBytes;
    if (p2.items && baseBytes + slotBytes <= 160 * 1024) ldsBytes += slotBytes;   // arg-max slots fit beside the |X| array
    else p2.items = nullptr;                                                         // very tall views: serial scan
    static size_t attrBytes = 0;
    if (attrBytes < ldsBytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&stftMapKernel<LR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsBytes));
        if (e != hipSuccess) return e;
        attrBytes = ldsBytes;
    }
    hipLaunchKernelGGL(stftMapKernel<LR>, dim3(grid), dim3(T), ldsBytes, stream, p2);
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
