// real_common.hpp -- helpers shared by the two forms of the channel-split K_A: spectrum_real.hip (R1 x 32 threads of 32 values) and
// spectrum_real16.hip (1024 threads of 16 values, N = 32768).  gfx950 only.
#pragma once
#include "chunk_map.hpp"
#include "fft_scalar.hpp"

namespace sgz {

// csf index -> LDS float index of this side's array: left holds csf[0 .. M], right csf[M .. N], both as entries 0 .. M at chunkPos()
struct ChannelIndex {
    int n, off;
    __device__ __forceinline__ int size() const { return n; }
    __device__ __forceinline__ int operator()(int k) const { return chunkPos(k - off); }
    __device__ __forceinline__ bool holds(int k) const { return k >= off && k <= off + n / 2; }
};

// floats of a side's magnitude array: entries 0 .. M at chunkPos(), then 16 zeroed floats (a tap window is read as kTapFloats
// contiguous floats from its first entry)
constexpr int realXFloats(int M) { return (chunkPos(M) + 1 + 16 + 1) & ~1; }

// floats behind the magnitudes that change hands during a workgroup's life: the pass-2 twiddle table (N >= 32768: 32 rows of 34 float2,
// 16-byte aligned: up to 2 floats of alignment in front), later the map's tile maxima (slots + 1) and chunk maxima (T)
constexpr int kTw2Row = 34 * 2, kTw2Floats = 32 * kTw2Row;
constexpr size_t realExtraFloats(uint32_t maxSlots, uint32_t T, bool tw2InLds)
{
    const size_t a = size_t(maxSlots) + 1 + T, b = tw2InLds ? kTw2Floats + 4 : 0;
    return a > b ? a : b;
}

// |X[k]| of the real-input transform from a = Z[k], b = Z[M - k] and w = W_N^k = (cos, -sin):
//   2 X = (a + conj b) - i w (a - conj b)
// The channel-split kernels transform x w / 2 -- the plan hands them the window halved (exact: a power of two) -- so that |2 X| of
// what they transform IS |X|: no multiply behind the root.
__device__ __forceinline__ float realBinMag(v2 a, v2 b, v2 w)
{
    const float ex = a.x + b.x, ey = a.y - b.y;          // a + conj b
    const float dx = a.x - b.x, dy = a.y + b.y;          // a - conj b
    // -i w d = -i (w.x + i w.y)(dx + i dy) = (w.x dy + w.y dx) + i (w.y dy - w.x dx)
    const float xr = ex + (w.x * dy + w.y * dx);
    const float xi = ey + (w.y * dy - w.x * dx);
    return __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
}

// Which (frame, pair, channel) a workgroup works on.  XCD-aware order (a speed assumption only): workgroup b runs on XCD b % 8, so inside
// each round of `roundSize` workgroups XCD x takes the x-th eighth of the round's units -- consecutive frames, whose 75 %-overlapping
// windows then share one L2.  32-bit arithmetic throughout (runStft refuses launches of 2^31 tasks): the 64-bit divisions this used to
// be written with were ~500 scalar instructions in front of every workgroup's first load.
struct UnitId { int side, pair; uint32_t task, frame, self; };
// (unit, nb: the workgroup's index and the launch's size -- or, for a workgroup that walks over several units, the index and the number
// of workgroups a one-unit-per-workgroup launch would have had)
// (P: RealParams, or the same block read in place from the kernel-argument segment -- WalkParams below)
template <bool MONO, typename P>
__device__ __forceinline__ UnitId unitOfIndex(const P &prm, uint32_t unit, const uint32_t nb)
{
    const uint32_t rs = prm.roundSize;
    if (nb >= 64u && rs >= 8u && (rs & 7u) == 0u) {
        const uint32_t base = (unit / rs) * rs;
        const uint32_t nbr = nb - base < rs ? nb - base : rs;
        const uint32_t x = (unit - base) & 7u, i = (unit - base) >> 3;
        const uint32_t per = nbr >> 3, extra = nbr & 7u;
        unit = base + x * per + (x < extra ? x : extra) + i;
    }
    UnitId u;
    u.side = MONO ? 0 : int(unit & 1u);
    uint32_t task = MONO ? unit : unit >> 1;                // (frame, pair), pair-major in the work list, frame-major in memory
    u.frame = task; u.pair = 0;
    if (prm.C > 1u) { const uint32_t F = uint32_t(prm.frames), pr = task / F, fr = task - pr * F; task = fr * prm.C + pr; u.frame = fr; u.pair = int(pr); }
    u.task = task;
    u.self = (task << 1) | uint32_t(u.side);                // ny / low / nyBest slots are indexed by task * 2 + side
    return u;
}

template <bool MONO>
__device__ __forceinline__ UnitId unitOf(const RealParams &prm) { return unitOfIndex<MONO>(prm, blockIdx.x, gridDim.x); }

// A kernel whose workgroups loop over units keeps every field of its parameter block it touches -- ~80 scalar registers -- alive across
// the loop, and what does not fit is parked in lanes of vector registers (v_writelane / v_readlane: 7 % of the N = 65536 kernel's
// vector time).  Such a kernel reads the block IN PLACE instead: the kernel-argument segment through a pointer that is opaque once per
// unit, so that every field is an s_load where it is used.  (The kernel's only parameter must be the block.)
typedef const RealParams __attribute__((address_space(4))) WalkParams;
__device__ __forceinline__ WalkParams *walkParams()
{
    WalkParams *p = (WalkParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
template <bool WALK> struct ParamsOf {
    typedef const RealParams T;
    static __device__ __forceinline__ T *get(const RealParams &launch) { return &launch; }
};
template <> struct ParamsOf<true> {
    typedef WalkParams T;
    static __device__ __forceinline__ T *get(const RealParams &) { return walkParams(); }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember what has been granted
inline hipError_t grantLds(const void *kernel, size_t need, size_t (&granted)[64])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && granted[dev] >= need) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(need));
    if (e == hipSuccess && dev >= 0 && dev < 64) granted[dev] = need;
    return e;
}

// LDS bytes of the 1024-thread form (spectrum_real16.hip): |X| / exchange areas, then the LDS-staged twiddle tables ([16][64] + [4][16] float2,
// 16-byte aligned) and column 0's scratch (128 floats) -- or, later in a workgroup's life, the map's tile and row maxima
constexpr size_t real16LdsBytes(uint32_t maxSlots)
{
    const size_t a = size_t((realXFloats(16384) + 3) & ~3) + (16 * 64 + 4 * 16) * 2 + 128, b = size_t(realXFloats(16384)) + maxSlots + 1 + 512;
    return (a > b ? a : b) * 4;
}
// spectrum_real16.hip: the 1024-thread form (N = 32768, pairs); binsIn: the bins-injection hook on that form's pixel map
hipError_t launchStftReal16(const RealParams &prm, hipStream_t stream);

}  // namespace sgz
