// spectrum_fft.hip -- K_A: fused  window x audio -> N-point complex FFT -> two-for-one split -> |X| ->
// pixel mapping (interpolate / arg-max) for one (frame, stereo pair) per workgroup.  gfx950 only.
//
// Replaces, per frame: TransformPair::prepareTransform (Source/Spectrum/TransformDSP.inl:39-231),
// doTransform (:487-502, cpl::dsp::UniFFT forward), and mapToLinearSpace (:506-1102) up to csp[].
//
// Structure (N = R^3, R = 32 for N = 32768, R = 16 for N = 4096; T = R^2 threads of R complex points each, all
// butterflies in VGPRs, LDS only for the two digit transposes, the csf array and the mapping):
//   pass 1  thread t        : R-point DIF over x[t + T j]  (coalesced strided L2 loads, window fused),
//                             times W_N^{t q}            -> exchange 1 (workgroup-wide, re then im)
//   pass 2  thread (q,t2)   : R-point DIF over j2,  times W_T^{t2 q2}  -> exchange 2 (inside R-lane groups)
//   pass 3  thread (q,q2)   : R-point DIF over t2  -> Z[q + R q2 + T m3]
//   mirror  : Z[k] meets Z[N-k] through one lane permutation per value (roles are laid out so that the mirror
//             partner of lane L is lane L ^ R) -> M[k] = |X1[k]|, M[N-k] = |X2[k]|  (csf of the reference after
//             :858-869), kept in LDS in a bank-padded natural order
//   mapping : plan.cpp's pixel records / arg-max pieces -> csp magnitude, written to HBM (8 KB / frame)
// HBM/L2 traffic per frame-pair: 2*W*4 B audio + W*4 B window + small L2-resident tables in, sides*P*4 B out.
// No MFMA: the path is fp32 butterflies + LDS transposes.
//
// Why R points per thread on T threads (16 waves, <= 128 VGPRs) rather than 2R points on T/2 threads: measured on MI355X
// (tools/ubench/valu.hip, valu2.hip, dif.hip -- whole-workgroup timing), a SIMD retires one plain fp32 VALU instruction
// per ~2.4 clocks, which two co-resident waves already saturate (one wave alone: one per ~4.8).  Four waves per SIMD do
// not add VALU throughput; what they add is cover for the load, LDS-exchange and barrier latencies of a frame whose
// phases are serialised by data dependencies.  Packed v_pk_*_f32 costs ~4.3 clocks per instruction (two plain ops: 4.8),
// v_sqrt 8.2, and v_cmp / v_cndmask / v_max / v_min / v_bfe or any instruction with an SGPR source ~4.1.
#include <atomic>

#include "stft_body.hpp"

namespace sgz {

template <int LR, int MIX, bool FULLW, bool WCOS = false>
__global__ void __launch_bounds__(1 << (2 * LR), LR == 4 ? 4 : 1)      // R = 16: four 256-thread workgroups per CU (<= 128 VGPRs)
stftMapKernel(const StftParams prm)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stftMapBody<LR, MIX, FULLW, -1, false, WCOS>(prm, lds, blockIdx.x, gridDim.x);
}

// load + window + three passes only: Z -> prm.zOut (Phase mode at N = R^3)
template <int LR, bool FULLW>
__global__ void __launch_bounds__(1 << (2 * LR))
stftComplexKernel(const StftParams prm)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stftMapBody<LR, 0, FULLW, -1, true>(prm, lds, blockIdx.x, gridDim.x);
}

template <int LR>
static hipError_t launchComplex(const StftParams &prm, int grid, hipStream_t stream);

// N = 2 R^3: workgroup b transforms half (b & 1) of task b >> 1 (stft_body.hpp, HALF)
template <int LR, int MIX, bool FULLW>
__global__ void __launch_bounds__(1 << (2 * LR))
stftHalfKernel(const StftParams prm)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    stftMapBody<LR, MIX, FULLW, 0>(prm, lds, blockIdx.x, gridDim.x);
    stftMapBody<LR, MIX, FULLW, 1>(prm, lds, blockIdx.x, gridDim.x);
}

// Pixel mapping of one side of one task from HBM-resident csf magnitudes (halves and generic paths): the csf range that side's
// records touch -- k in [N-15, N] + [0, N/2+31] on the left, [N/2-16, N] + [0, 30] on the right (plan.cpp checks it) --
// is staged in LDS and mapped by the fused kernel's balanced scan.  NT threads per (task, side) unit, 1024 / NT units per
// workgroup, each with its own slice of the LDS (`unitFloats` apart): a small transform has thousands of units of ~1 pixel
// per thread at NT = 1024, whose cost is the fixed part (table reads, barrier) -- NT = 256 shares it between four units.
template <int NT>
__global__ void __launch_bounds__(1024)
mapSideKernel(const StftParams prm, const float *bins, const uint32_t N, float *mapped, const long units, const uint32_t unitFloats)
{
    extern __shared__ __attribute__((aligned(16))) float ldsAll[];
    const int tid = threadIdx.x % NT;
    const long unit = long(blockIdx.x) * (1024 / NT) + threadIdx.x / NT;
    const bool live = unit < units;                                     // (a dead unit of the last workgroup redoes the last one)
    const long u = live ? unit : units - 1;
    float *lds = ldsAll + size_t(threadIdx.x / NT) * unitFloats;
    const long task = u / prm.sides;
    const int side = int(u - task * prm.sides);
    const int half = int(N >> 1), count = half + 48;
    const OneSideIndex at{int(N), side ? half + 17 : 16};
    const uint32_t nLeft = prm.nItemsLeft, nSide = side ? prm.nItems - nLeft : nLeft;
    const MapView v{prm.items + (side ? nLeft : 0u), nSide, side ? 0u : nSide, side ? nLeft : 0u, prm.recs + side * prm.P, int(prm.P),
                    side ? 0 : int(prm.P), mapped + (size_t(task) * prm.sides + side) * prm.P};
    float *win = lds + ((count + (count >> 5) + 2) & ~1);
    MapPixelsBalanced<5, NT, OneSideIndex> mapper;
    SGZ_CLK_HALF(side == int((prm.ablate >> 15) & 1u));       // debug clocks: which side reports
    SGZ_CLK(0);
    mapper.prefetchTables(v, tid);
    const float *src = bins + size_t(task) * (size_t(N) + 1);
    const bool split = prm.binsSplit != 0;
    // split: csf as the two half-frame workgroups wrote it -- even bins [0, N/2] (csf[N] last), then the odd bins
    // LB independent loads in flight per thread (N = 65536: 33 elements per thread = two round trips; N = 8192: one of 5)
    auto stage = [&](auto lb) {
        constexpr int LB = decltype(lb)::value;
        for (int i0 = tid; i0 < count; i0 += LB * NT) {
            float val[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int i = i0 + u * NT;
                int k = i - at.off;
                k = k < 0 ? k + int(N) + 1 : k;
                k = k > int(N) ? int(N) : k;                           // (past the end: any valid address, the value is dropped)
                const int e = !split ? k : ((k & 1) ? half + 1 + (k >> 1) : (k >> 1));
                val[u] = src[e];
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int i = i0 + u * NT;
                if (i < count) lds[i + (i >> 5)] = val[u];
            }
        }
    };
    if (count <= 5 * NT) stage(std::integral_constant<int, 5>{});
    else stage(std::integral_constant<int, 17>{});
    mapper.prefetchWeights(prm, tid, v.total);
    ldsBarrier();
    SGZ_CLK(7);
    mapper.run(prm, v, at, lds, win, tid, task);
    SGZ_CLK(9);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember, per device, how much each
// kernel has been granted (atomics: launches may come from several host threads, one process may drive several devices).
struct LdsGrant {
    static constexpr int kDevices = 64;
    std::atomic<size_t> bytes[kDevices];
    hipError_t ensure(const void *kernel, size_t need)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const bool cached = dev >= 0 && dev < kDevices;
        if (cached && bytes[dev].load(std::memory_order_acquire) >= need) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(need));
        if (e == hipSuccess && cached) bytes[dev].store(need, std::memory_order_release);
        return e;
    }
};

static size_t mapSidesLds(const StftParams &prm, uint32_t N)          // bytes of one unit
{
    const int count = int(N / 2) + 48;
    const uint32_t maxSide = prm.nItemsLeft > prm.nItems - prm.nItemsLeft ? prm.nItemsLeft : prm.nItems - prm.nItemsLeft;
    return size_t((count + (count >> 5) + 2) & ~1) * sizeof(float) + size_t(maxSide) * 4;
}
bool mapSidesFit(const StftParams &prm, uint32_t N) { return mapSidesLds(prm, N) <= 160 * 1024; }

hipError_t launchMapSides(const StftParams &prm, uint32_t N, const float *bins, long ntasks, float *mapped, hipStream_t stream)
{
    const size_t ldsBytes = mapSidesLds(prm, N);
    if (ldsBytes > 160 * 1024) return hipErrorInvalidValue;
    const long units = ntasks * long(prm.sides);
    const uint32_t unitFloats = uint32_t((ldsBytes + 15) / 16 * 4);     // 16-byte aligned slices
    static LdsGrant grant[2];
    if (N <= 8192 && size_t(unitFloats) * 4 * 4 <= 160 * 1024) {        // four 256-thread units per workgroup
        const size_t bytes = size_t(unitFloats) * 4 * 4;
        if (hipError_t e = grant[0].ensure(reinterpret_cast<const void *>(&mapSideKernel<256>), bytes); e != hipSuccess) return e;
        hipLaunchKernelGGL(mapSideKernel<256>, dim3(unsigned((units + 3) / 4)), dim3(1024), bytes, stream, prm, bins, N, mapped, units,
                           unitFloats);
    } else {
        if (hipError_t e = grant[1].ensure(reinterpret_cast<const void *>(&mapSideKernel<1024>), ldsBytes); e != hipSuccess) return e;
        hipLaunchKernelGGL(mapSideKernel<1024>, dim3(unsigned(units)), dim3(1024), ldsBytes, stream, prm, bins, N, mapped, units, unitFloats);
    }
    return hipGetLastError();
}

template <int LR>
static hipError_t launchHalves(const StftParams &prm, int grid, hipStream_t stream)
{
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const size_t ldsBytes = (size_t(N) + (N >> LR) + 12 + 2 * R + 4 + 2 * kSpecBins) * sizeof(float);
    const bool simple = prm.mode == SGZ_CH_SEPARATE || prm.mode == SGZ_CH_COMPLEX;
    const bool fullw = prm.W == uint32_t(2 * N);
    using Kern = void (*)(const StftParams);
    static const Kern kerns[4] = {&stftHalfKernel<LR, 1, false>, &stftHalfKernel<LR, 1, true>, &stftHalfKernel<LR, 0, false>,
                                  &stftHalfKernel<LR, 0, true>};
    const int which = (simple ? 2 : 0) + (fullw ? 1 : 0);
    static LdsGrant grant[4];
    if (hipError_t e = grant[which].ensure(reinterpret_cast<const void *>(kerns[which]), ldsBytes); e != hipSuccess) return e;
    StftParams p2 = prm;
    p2.items = nullptr;
    hipLaunchKernelGGL(kerns[which], dim3(grid), dim3(T), ldsBytes, stream, p2);
    return hipGetLastError();
}

hipError_t launchStftHalves(const StftParams &prm, uint32_t N, int grid, hipStream_t stream)
{
    switch (N) {
    case 65536: return launchHalves<5>(prm, grid, stream);
    case 8192: return launchHalves<4>(prm, grid, stream);
    default: return hipErrorNotSupported;
    }
}

template <int LR>
static hipError_t launchStft(const StftParams &prm, int grid, hipStream_t stream)
{
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const size_t baseBytes = (size_t(N) + (N >> LR) + 12 + 2 * R + 4 + 2 * kSpecBins) * sizeof(float);
    const size_t slotBytes = size_t(prm.nItems) * 4;
    StftParams p2 = prm;
    size_t ldsBytes = baseBytes;
    if (p2.items && baseBytes + slotBytes <= 160 * 1024) ldsBytes += slotBytes;   // arg-max slots fit beside the |X| array
    else p2.items = nullptr;                                                         // very tall views: serial scan
    const int mix = prm.mode == SGZ_CH_SEPARATE ? 0 : (prm.mode == SGZ_CH_COMPLEX ? 2 : 1);
    const bool fullw = prm.W == uint32_t(N);
    using Kern = void (*)(const StftParams);
    // (the last three: full window evaluated in the kernel -- Hann / Hamming periodic, Plan::winPhaseT)
    static const Kern kerns[9] = {&stftMapKernel<LR, 0, false>, &stftMapKernel<LR, 0, true>, &stftMapKernel<LR, 1, false>,
                                  &stftMapKernel<LR, 1, true>,  &stftMapKernel<LR, 2, false>, &stftMapKernel<LR, 2, true>,
                                  &stftMapKernel<LR, 0, true, true>, &stftMapKernel<LR, 1, true, true>, &stftMapKernel<LR, 2, true, true>};
    const int which = (fullw && prm.winPhase) ? 6 + mix : 2 * mix + (fullw ? 1 : 0);
    static LdsGrant grant[9];
    if (hipError_t e = grant[which].ensure(reinterpret_cast<const void *>(kerns[which]), ldsBytes); e != hipSuccess) return e;
    hipLaunchKernelGGL(kerns[which], dim3(grid), dim3(T), ldsBytes, stream, p2);
    return hipGetLastError();
}

template <int LR>
static hipError_t launchComplex(const StftParams &prm, int grid, hipStream_t stream)
{
    constexpr int R = 1 << LR, T = R * R, N = R * T;
    const size_t ldsBytes = (size_t(N) + (N >> LR) + 12 + 2 * R + 4 + 2 * kSpecBins) * sizeof(float);
    const bool fullw = prm.W == uint32_t(N);
    using Kern = void (*)(const StftParams);
    static const Kern kerns[2] = {&stftComplexKernel<LR, false>, &stftComplexKernel<LR, true>};
    static LdsGrant grant[2];
    if (hipError_t e = grant[fullw].ensure(reinterpret_cast<const void *>(kerns[fullw]), ldsBytes); e != hipSuccess) return e;
    hipLaunchKernelGGL(kerns[fullw], dim3(grid), dim3(T), ldsBytes, stream, prm);
    return hipGetLastError();
}

hipError_t launchStftComplex(const StftParams &prm, uint32_t N, int grid, hipStream_t stream)
{
    switch (N) {
    case 32768: return launchComplex<5>(prm, grid, stream);
    case 4096: return launchComplex<4>(prm, grid, stream);
    default: return hipErrorNotSupported;
    }
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
