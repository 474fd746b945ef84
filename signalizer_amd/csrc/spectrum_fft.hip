// spectrum_fft.hip -- K_A: fused  window x audio -> N-point complex FFT -> two-for-one split -> |X| ->
// pixel mapping (interpolate / arg-max) for one (frame, stereo pair) per workgroup.  gfx950 only.
//
// Replaces, per frame: TransformPair::prepareTransform (Source/Spectrum/TransformDSP.inl:39-231),
// doTransform (:487-502, cpl::dsp::UniFFT forward), and mapToLinearSpace (:506-1102) up to csp[].
//
// Structure (N = R^3, R = 32 for N = 32768, R = 16 for N = 4096; T = R^2 virtual threads of R points, two per
// real thread (see stftMapKernel),
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

#include <type_traits>

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

#define SGZ_CLK(slot)                                                                                   \
    do {                                                                                                \
        if (prm.phaseClock && tid == 0 && task == long(prm.ablate >> 16))                   \
            prm.phaseClock[slot] = __builtin_readcyclecounter();                                         \
    } while (0)

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

// balanced version; `win` = one uint32 per MaxItem in LDS.  Contains one workgroup barrier and NO atomics
// (ds_max_u64 runs at only ~1 lane-op per 4-6 cycles: 3400 of them cost ~13k cycles per frame).
//   (a) every <=16-bin piece finds its local winner (first strictly greater |X|^2) and stores the winner's scan offset;
//   (c) the pixel's thread replays the reference's scan over its pieces' winners, in order -> same arg-max, same ties.
// Table reads (items, records, tap weights) are issued in batches of independent loads: a thread's work list is
// tiny, so what matters is the number of dependent global-load round trips, not the byte count.
template <int LR, int NT>
__device__ __forceinline__ void mapPixelsBalanced(const StftParams &prm, const float *lds, uint32_t *win, int tid, long task)
{
#pragma clang fp contract(off)
    constexpr int R = 1 << LR, N = R * R * R;
    constexpr int IB = 8;                                                // items per thread per batch
    constexpr int RB = 4;                                                // records per thread per batch
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const int total = int(prm.sides * prm.P);
    float *out = prm.mapped + size_t(task) * total;
    // (a) arg-max pieces.  A piece is a 16-aligned window of csf (one 32-block of the padded layout, so its 16 floats
    // are contiguous: one base address + immediate offsets) with positions lo..hi valid; values outside are ANDed
    // to +0, which can never be "strictly greater".  Left-side pieces scan k upwards, right-side pieces downwards.
    auto scanPieces = [&](uint32_t first, uint32_t last, auto rightSide) {
        constexpr bool RIGHT = decltype(rightSide)::value;
        for (uint32_t base = first; base < last; base += NT * IB) {
            MaxItem item[IB];
#pragma unroll
            for (int b = 0; b < IB; ++b) {
                const uint32_t it = base + b * NT + tid;
                item[b] = it < last ? prm.items[it] : MaxItem{0u, 0u};
            }
            float mv[IB][16];
#pragma unroll
            for (int b = 0; b < IB; ++b) {
                const int k0 = int(item[b].win & 0xFFFFu) << 4;
                const float *src = lds + (k0 + (k0 >> LR));
#pragma unroll
                for (int j = 0; j < 16; ++j) mv[b][j] = src[j];
            }
            uint32_t winner[IB];
#pragma unroll
            for (int b = 0; b < IB; ++b) {
                const int k0 = int(item[b].win & 0xFFFFu) << 4;
                const int lo = int((item[b].win >> 16) & 15u), hi = int((item[b].win >> 20) & 15u);
                // valid-position mask: bits lo..hi
                const uint32_t mask = (0xFFFFu >> (15 - hi)) & (0xFFFFu << lo);
                float best = 0.f;
                uint32_t bestK = kNone;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int j = RIGHT ? 15 - jj : jj;               // scan order = ascending offset
                    const float sq = mv[b][j] * mv[b][j];             // Math::square(csf[offset]) with imag == 0 (x*x + 0 == x*x)
                    const uint32_t keep = uint32_t(__builtin_amdgcn_sbfe(int(mask), j, 1));   // 0 or ~0
                    const float sqm = __uint_as_float(__float_as_uint(sq) & keep);
                    const bool take = sqm > best;                     // first strictly greater wins (TransformDSP.inl:965)
                    best = take ? sqm : best;
                    bestK = take ? uint32_t(k0 + j) : bestK;
                }
                winner[b] = bestK;
            }
#pragma unroll
            for (int b = 0; b < IB; ++b) {
                const uint32_t it = base + b * NT + tid;
                if (it < last) win[it] = winner[b];
            }
        }
    };
    scanPieces(0u, prm.nItemsLeft, std::false_type{});
    scanPieces(prm.nItemsLeft, prm.nItems, std::true_type{});
    SGZ_CLK(10);
    // (b) interpolated pixels (<= 10 taps, accumulated in tap order)
    const bool oneBatch = total <= NT * RB;                              // then the records stay in registers across the barrier
    PixelRec rec[RB];
    for (int base = 0; base < total; base += NT * RB) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int idx = base + b * NT + tid;
            rec[b] = idx < total ? prm.recs[idx] : PixelRec{2, 0, 0, 0};
        }
        // tap weights and magnitudes: unconditional, independent loads (the weight table is padded by kMaxTaps zeros)
        float w[RB][kMaxTaps], mv[RB][kMaxTaps];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int wbase = rec[b].kind == 0 ? rec[b].c : 0;
            int k = rec[b].kind == 0 ? rec[b].a : 0;
#pragma unroll
            for (int i = 0; i < kMaxTaps; ++i) {
                w[b][i] = prm.weights[wbase + i];
                mv[b][i] = lds[k + (k >> LR)];
                k = (k == N) ? 0 : k + 1;
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int idx = base + b * NT + tid;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < kMaxTaps; ++i) {
                const float prod = mv[b][i] * w[b][i];
                acc = (i < rec[b].b) ? acc + prod : acc;               // taps accumulate in order (lanczosFilter restatement)
            }
            if (rec[b].kind == 0) out[idx] = finishPixel<LR>(prm.invSize * acc);
        }
    }
    SGZ_CLK(11);
    __syncthreads();
    SGZ_CLK(12);
    // (c) resolve the arg-max pixels from their pieces' winners
    for (int base = 0; base < total; base += NT * RB) {
        if (!oneBatch) {
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int idx = base + b * NT + tid;
                rec[b] = idx < total ? prm.recs[idx] : PixelRec{2, 0, 0, 0};
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            if ((rec[b].kind & 1) == 0) continue;
            const int idx = base + b * NT + tid;
            const bool right = idx >= int(prm.P);
            const int first = rec[b].kind >> 1;
            // number of 16-aligned csf windows the run [a, a+b) spans (k = offset, or N - offset on the right side)
            const int kLo = right ? N - (rec[b].a + rec[b].b - 1) : rec[b].a;
            const int kHi = right ? N - rec[b].a : rec[b].a + rec[b].b - 1;
            const int pieces = (kHi >> 4) - (kLo >> 4) + 1;
            float best = 0.f;
            int k = rec[b].c;                                            // maxLBin = maxRBin = bin (TransformDSP.inl:953)
            for (int pc = 0; pc < pieces; ++pc) {
                const uint32_t kk = win[first + pc];
                if (kk != kNone) {
                    const float m = lds[kk + (kk >> LR)];
                    const float sq = m * m + 0.f;
                    if (sq > best) { best = sq; k = int(kk); }
                }
            }
            out[idx] = finishPixel<LR>(prm.invSize * lds[k + (k >> LR)]);
        }
    }
}

// An opaque copy of a per-thread constant: address arithmetic derived from it cannot be hoisted out of the frame
// loop (where it would pin VGPRs for the whole iteration and spill); it is recomputed where it is used instead.
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}


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

// One workgroup = one (frame, pair) at a time.  LR = log2(R), N = R^3, T = R^2 "virtual threads" of R points.
//
// A REAL thread owns TWO virtual threads (sets A and B): TR = T/2 real threads, 2R complex points = 4R data VGPRs,
// 8 waves per workgroup at R = 32 -> a 256-VGPR budget per thread, i.e. no spills and room to keep the next
// frame's samples in flight.  Roles:
//   pass 1    : columns tA = tid, tB = tid + TR                       (x[t + T j], j < R)
//   passes 2/3: group p = tid / R, lane l = tid % R.   A = (q = p,           t2|q2 = l)
//                                                      B = (q = R-p (R/2 if p = 0), t2|q2 = R-1-l)
// A and B of one thread are mirror partners: bin k = q + R q2 + T m3 of A pairs with N-k = B's bin R-1-m3, so the
// two-for-one split (TransformDSP.inl:858) is pure register arithmetic.  Only group p = 0 (q = 0 and q = R/2, which
// mirror onto themselves) needs a lane permutation (ds_bpermute inside its half-wave).
template <int LR>
__global__ void __launch_bounds__((1 << (2 * LR)) / 2)
stftMapKernel(const StftParams prm)
{
    constexpr int R = 1 << LR;
    constexpr int T = R * R;
    constexpr int TR = T / 2;
    constexpr int N = R * T;
    constexpr int PADSTRIDE = T + (T >> LR);          // padded distance between k and k + T
    constexpr int SCRATCH = N + (N >> LR) + 4;        // float index of column 0's 2R-float scratch
    constexpr int SLOTS = SCRATCH + 2 * R + 4;        // float index of the arg-max piece winners (nItems u32)
    constexpr int TILE = R * (R + 1);
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const long tasks = prm.frames * long(prm.C);
    const int p = tid >> LR, l = tid & (R - 1), lb = R - 1 - l;
    const int qA = p, qB = (p == 0) ? R / 2 : R - p;
    const bool split = (prm.sides == 2);
    const int mode = prm.mode;
    uint32_t *win = reinterpret_cast<uint32_t *>(lds + SLOTS);          // one winner offset per arg-max piece
    const bool balanced = prm.items != nullptr;
    // prepareTransform channel mixes (TransformDSP.inl:59-216): re = (mixRL*L + mixRR*R)*w*mixS, im likewise
    float mixRL = 1.f, mixRR = 0.f, mixIL = 0.f, mixIR = 1.f, mixS = 1.f;      // Phase / Separate / Complex
    if (mode == SGZ_CH_LEFT) { mixIR = 0.f; }
    else if (mode == SGZ_CH_RIGHT) { mixRL = 0.f; mixRR = 1.f; mixIR = 0.f; }
    else if (mode == SGZ_CH_MERGE) { mixRR = 1.f; mixIR = 0.f; mixS = 0.5f; }
    else if (mode == SGZ_CH_SIDE) { mixRR = -1.f; mixIR = 0.f; mixS = 0.5f; }
    else if (mode == SGZ_CH_MIDSIDE) { mixRR = 1.f; mixIL = 1.f; mixIR = -1.f; mixS = 0.5f; }

    float reA[R], imA[R], reB[R], imB[R];

    // raw samples of one column: L -> re, R -> im (strided dword buffer loads; reads past W return 0 = zero padding)
    auto issueAudio = [&](long task) {
        const long frame = task / prm.C;
        const int pair = int(task - frame * prm.C);
        const float *L = prm.planar + size_t(2 * pair) * prm.chStride + size_t(frame) * prm.hop;
        const __amdgpu_buffer_rsrc_t rsL = makeRsrc(L, prm.W * 4u);
        const __amdgpu_buffer_rsrc_t rsR = makeRsrc(L + prm.chStride, prm.W * 4u);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            reA[j] = bufLoad(rsL, tid * 4, j * (T * 4));
            imA[j] = bufLoad(rsR, tid * 4, j * (T * 4));
            reB[j] = bufLoad(rsL, (tid + TR) * 4, j * (T * 4));
            imB[j] = bufLoad(rsR, (tid + TR) * 4, j * (T * 4));
        }
    };
    // window the samples (prepareTransform); the window table is L2 resident
    auto applyWindow = [&]() {
        const __amdgpu_buffer_rsrc_t rsW = makeRsrc(prm.window, prm.W * 4u);
        constexpr int WB = R;                                          // all window loads in flight together
#pragma unroll
        for (int jb = 0; jb < R; jb += WB) {
            float wa[WB], wb[WB];
#pragma unroll
            for (int j = 0; j < WB; ++j) {
                wa[j] = bufLoad(rsW, tid * 4, (jb + j) * (T * 4));
                wb[j] = bufLoad(rsW, (tid + TR) * 4, (jb + j) * (T * 4));
            }
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) {
                const int j = jb + jj;
                // branch-free channel mix: (a*l + b*r) * w * s with a, b in {0, +-1}, s in {1, 0.5} rounds exactly
                // like the reference's `(l +- r) * w * 0.5f` / `l * w` (adding a signed zero is exact)
                const float la = reA[j], ra = imA[j], lb_ = reB[j], rb = imB[j];
                reA[j] = (mixRL * la + mixRR * ra) * wa[jj] * mixS;
                imA[j] = (mixIL * la + mixIR * ra) * wa[jj] * mixS;
                reB[j] = (mixRL * lb_ + mixRR * rb) * wb[jj] * mixS;
                imB[j] = (mixIL * lb_ + mixIR * rb) * wb[jj] * mixS;
            }
        }
    };

    // One workgroup per task, no persistent frame loop: with a loop, LLVM hoists dozens of loop-invariant LDS/global
    // address computations out of it and immediately spills them (measured: 33 prologue spills, ~50 reloads on the
    // critical path of every frame).  Straight-line code has no such hoisting, and the dispatcher refills CUs anyway.
    //
    // XCD-aware task order.  Workgroup b is observed to run on XCD b % 8 (a speed assumption only, never a
    // correctness one): XCD x gets the contiguous task range [base(x), base(x+1)), so that the workgroups sharing an
    // L2 walk consecutive (75 %-overlapping) frames together and each sample is fetched from HBM once per XCD.
    long task = blockIdx.x;
    if (gridDim.x % 8 == 0 || tasks >= 64) {
        const long nb = gridDim.x, x = blockIdx.x % 8, i = blockIdx.x / 8;
        const long per = nb / 8, extra = nb % 8;                   // XCD x owns per + (x < extra) workgroups
        task = x * per + (x < extra ? x : extra) + i;
    }
    {
        SGZ_CLK(0);
        if (prm.binsIn == nullptr) {
            issueAudio(task);
            applyWindow();
            // ------------------------------------------------------------------ pass 1: DIF (samples already windowed)
            {
                if (!(prm.ablate & 1)) { dif<R, R, 0>(reA, imA); dif<R, R, 0>(reB, imB); }
                if (!(prm.ablate & 32)) {
                    const __amdgpu_buffer_rsrc_t rs = makeRsrc(prm.tw1, uint32_t(3 + R / 4 - 1) * T * 8u);
                    TwFactors<LR> ta, tb;
                    ta.load(rs, tid * 8, T * 8);
                    tb.load(rs, (tid + TR) * 8, T * 8);
                    ta.apply(reA, imA);                                // times W_N^{t q}
                    tb.apply(reB, imB);
                }
            }
            SGZ_CLK(1);
            // ---------------------------------------------------------- exchange 1 (workgroup-wide; re then im)
            __syncthreads();                                           // previous frame's mapping reads are done
            if (!(prm.ablate & 2)) {
                const int rdA = qA * T + l, rdB = qB * T + lb;
#pragma unroll
                for (int qq = 0; qq < R; ++qq) { lds[qq * T + tid] = reA[brev(qq, LR)]; lds[qq * T + tid + TR] = reB[brev(qq, LR)]; }
                __syncthreads();
#pragma unroll
                for (int j2 = 0; j2 < R; ++j2) { reA[j2] = lds[rdA + R * j2]; reB[j2] = lds[rdB + R * j2]; }
                __syncthreads();
#pragma unroll
                for (int qq = 0; qq < R; ++qq) { lds[qq * T + tid] = imA[brev(qq, LR)]; lds[qq * T + tid + TR] = imB[brev(qq, LR)]; }
                __syncthreads();
#pragma unroll
                for (int j2 = 0; j2 < R; ++j2) { imA[j2] = lds[rdA + R * j2]; imB[j2] = lds[rdB + R * j2]; }
            }
            SGZ_CLK(2);
            // ------------------------------------------------------------------ pass 2 (A: t2 = l, B: t2 = R-1-l)
            {
                if (!(prm.ablate & 1)) { dif<R, R, 0>(reA, imA); dif<R, R, 0>(reB, imB); }
                if (!(prm.ablate & 32)) {
                    const __amdgpu_buffer_rsrc_t rs = makeRsrc(prm.tw2, uint32_t(3 + R / 4 - 1) * R * 8u);
                    TwFactors<LR> ta, tb;
                    ta.load(rs, l * 8, R * 8);
                    tb.load(rs, lb * 8, R * 8);
                    ta.apply(reA, imA);                                // times W_T^{t2 q2}
                    tb.apply(reB, imB);
                }
            }
            SGZ_CLK(3);
            // ------------------------------- exchange 2: R x R transposes inside each R-lane group (wave-local tiles)
            __syncthreads();                                           // every wave has finished reading exchange 1
            if (!(prm.ablate & 4)) {
                const int tA = p * TILE, tB = (R / 2 + p) * TILE;
#pragma unroll
                for (int q2 = 0; q2 < R; ++q2) { lds[tA + q2 * (R + 1) + l] = reA[brev(q2, LR)]; lds[tB + q2 * (R + 1) + lb] = reB[brev(q2, LR)]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int j = 0; j < R; ++j) { reA[j] = lds[tA + l * (R + 1) + j]; reB[j] = lds[tB + lb * (R + 1) + j]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q2 = 0; q2 < R; ++q2) { lds[tA + q2 * (R + 1) + l] = imA[brev(q2, LR)]; lds[tB + q2 * (R + 1) + lb] = imB[brev(q2, LR)]; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int j = 0; j < R; ++j) { imA[j] = lds[tA + l * (R + 1) + j]; imB[j] = lds[tB + lb * (R + 1) + j]; }
            }
            SGZ_CLK(4);
            // ------------------------------------------------------------------ pass 3 (A: q2 = l, B: q2 = R-1-l)
            if (!(prm.ablate & 1)) { dif<R, R, 0>(reA, imA); dif<R, R, 0>(reB, imB); }
            SGZ_CLK(5);
            // X[c + T m3] at register brev(m3);  cA = qA + R l,  cB = qB + R (R-1-l)
            const int cA = qA + R * l, cB = qB + R * lb;
            const int baseA = cA + (cA >> LR), baseB = cB + (cB >> LR);    // padded LDS address of k = c
            if (split && !(prm.ablate & 8)) {
                if (tid == 0) {                                        // column 0 (q = 0, q2 = 0) mirrors onto itself: redone below
#pragma unroll
                    for (int m3 = 0; m3 < R; ++m3) {
                        lds[SCRATCH + 2 * m3] = reA[brev(m3, LR)];
                        lds[SCRATCH + 2 * m3 + 1] = imA[brev(m3, LR)];
                    }
                }
                // mirror lanes for group p = 0 (inside lanes 0..R-1 of wave 0): A (q = 0): q2' = R - q2 ; B (q = R/2): lane R-1-l
                const int laneA0 = (((R - l) & (R - 1))) << 2, laneB0 = (R - 1 - l) << 2;
#pragma unroll
                for (int m3 = 0; m3 < R / 2; ++m3) {
                    const int ia = brev(m3, LR), ib = brev(R - 1 - m3, LR);      // k < N/2 at ia, k > N/2 at ib
                    // mirror values: Z[N - k] for the four bins (A,ia) (A,ib) (B,ia) (B,ib)
                    float mAa_r = reB[ib], mAa_i = imB[ib], mAb_r = reB[ia], mAb_i = imB[ia];
                    float mBa_r = reA[ib], mBa_i = imA[ib], mBb_r = reA[ia], mBb_i = imA[ia];
                    if (p == 0) {
                        mAa_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneA0, __builtin_bit_cast(int, reA[ib])));
                        mAa_i = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneA0, __builtin_bit_cast(int, imA[ib])));
                        mAb_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneA0, __builtin_bit_cast(int, reA[ia])));
                        mAb_i = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneA0, __builtin_bit_cast(int, imA[ia])));
                        mBa_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneB0, __builtin_bit_cast(int, reB[ib])));
                        mBa_i = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneB0, __builtin_bit_cast(int, imB[ib])));
                        mBb_r = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneB0, __builtin_bit_cast(int, reB[ia])));
                        mBb_i = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(laneB0, __builtin_bit_cast(int, imB[ia])));
                    }
                    // k < N/2: X1 = (Z[k] + conj Z[N-k])/2 ; k > N/2: X2 = (Z[N-k] - conj Z[k])/(2i)  (magnitudes only)
                    const float uAa = reA[ia] + mAa_r, vAa = imA[ia] - mAa_i, uAb = reA[ib] - mAb_r, vAb = imA[ib] + mAb_i;
                    const float uBa = reB[ia] + mBa_r, vBa = imB[ia] - mBa_i, uBb = reB[ib] - mBb_r, vBb = imB[ib] + mBb_i;
                    reA[ia] = 0.5f * __builtin_amdgcn_sqrtf(uAa * uAa + vAa * vAa);
                    reA[ib] = 0.5f * __builtin_amdgcn_sqrtf(uAb * uAb + vAb * vAb);
                    reB[ia] = 0.5f * __builtin_amdgcn_sqrtf(uBa * uBa + vBa * vBa);
                    reB[ib] = 0.5f * __builtin_amdgcn_sqrtf(uBb * uBb + vBb * vBb);
                }
            } else {
                if (tid == 0) { lds[SCRATCH] = reA[0]; lds[SCRATCH + 1] = imA[0];
                                lds[SCRATCH + R] = reA[brev(R / 2, LR)]; lds[SCRATCH + R + 1] = imA[brev(R / 2, LR)]; }
#pragma unroll
                for (int m3 = 0; m3 < R; ++m3) {                       // csf[k] = |Z[k]| (TransformDSP.inl:553-560, :993-1002)
                    const int i = brev(m3, LR);
                    reA[i] = __builtin_amdgcn_sqrtf(reA[i] * reA[i] + imA[i] * imA[i]);
                    reB[i] = __builtin_amdgcn_sqrtf(reB[i] * reB[i] + imB[i] * imB[i]);
                }
            }
            SGZ_CLK(6);
            __syncthreads();                                           // exchange-2 tiles are dead: M may overwrite them
#pragma unroll
            for (int m3 = 0; m3 < R; ++m3) {
                lds[baseA + m3 * PADSTRIDE] = reA[brev(m3, LR)];
                lds[baseB + m3 * PADSTRIDE] = reB[brev(m3, LR)];
            }
            __syncthreads();
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
            if (split && tid == R) {
                const int kq = N / 2 - 1;
                lds[kq + (kq >> LR)] *= 0.5f;                        // csf[N/2-1] *= 0.5 (quirk Q3, :864)
            }
            __syncthreads();
        } else {
            // test path (sgz_stage_map_from_bins): csf magnitudes come from HBM
            const float *src = prm.binsIn + size_t(task) * (N + 1);
            __syncthreads();
            for (int k = tid; k <= N; k += TR) lds[k + (k >> LR)] = src[k];
            __syncthreads();
        }
        SGZ_CLK(7);

        if (prm.binsOut) {
            float *dst = prm.binsOut + size_t(task) * (N + 1);
            for (int k = tid; k <= N; k += TR) dst[k] = lds[k + (k >> LR)];
        }
        SGZ_CLK(8);
        // ---------------------------------------------------------------------- pixel mapping
        if (prm.mapped && !(prm.ablate & 16)) {
            if (balanced) mapPixelsBalanced<LR, TR>(prm, lds, win, tid, task);
            else mapPixelsSerial<LR, TR>(prm, lds, tid, task);
        }
        SGZ_CLK(9);
    }
}

template <int LR>
static hipError_t launchStft(const StftParams &prm, int grid, hipStream_t stream)
{
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const size_t baseBytes = (size_t(N) + (N >> LR) + 4 + 2 * R + 4) * sizeof(float);
    const size_t slotBytes = size_t(prm.nItems) * 4;
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
    hipLaunchKernelGGL(stftMapKernel<LR>, dim3(grid), dim3(T / 2), ldsBytes, stream, p2);
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
