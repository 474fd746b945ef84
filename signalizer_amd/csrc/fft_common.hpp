// fft_common.hpp -- device helpers shared by the fused FFT kernels (spectrum_fft.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgz {

typedef float v2 __attribute__((ext_vector_type(2)));      // an aligned VGPR pair: (re, im) of one complex value

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

// W_64^j = cos64(j) - i sin64(j), j = 0..32: the pre-rotation of the odd half of a 2 R^3-point frame (stft_body.hpp, HALF = 1)
__host__ __device__ constexpr float cos64(int j)
{
    constexpr float v[33] = {1.0f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f,
                             0.92387953251128675613f, 0.88192126434835502971f, 0.83146961230254523708f,
                             0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f,
                             0.55557023301960222474f, 0.47139673682599764856f, 0.38268343236508977173f,
                             0.29028467725446236764f, 0.19509032201612826785f, 0.098017140329560601994f, 0.0f,
                             -0.098017140329560601994f, -0.19509032201612826785f, -0.29028467725446236764f,
                             -0.38268343236508977173f, -0.47139673682599764856f, -0.55557023301960222474f,
                             -0.63439328416364549822f, -0.70710678118654752440f, -0.77301045336273696081f,
                             -0.83146961230254523708f, -0.88192126434835502971f, -0.92387953251128675613f,
                             -0.95694033573220886494f, -0.98078528040323044913f, -0.99518472667219688624f, -1.0f};
    return v[j];
}
__host__ __device__ constexpr float sin64(int j) { return j <= 16 ? cos64(16 - j) : cos64(j - 16); }

__host__ __device__ constexpr int brev(int x, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
    return r;
}


// In-register radix-2 DIF over LEN elements starting at BASE; result is in bit-reversed order.  V = float or v2.
template <typename V, int R, int LEN, int BASE>
__device__ __forceinline__ void dif(V (&re)[R], V (&im)[R])
{
    constexpr int H = LEN / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int a = BASE + i, b = BASE + i + H;
        const V ar = re[a], ai = im[a], br = re[b], bi = im[b];
        re[a] = ar + br;
        im[a] = ai + bi;
        const V dr = ar - br, di = ai - bi;
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
        dif<V, R, H, BASE>(re, im);
        dif<V, R, H, BASE + H>(re, im);
    }
}

// ---- packed complex arithmetic: one value = (re, im) in an aligned VGPR pair, VOP3P ops with op_sel / neg modifiers.
// Measured (tools/ubench/valu.hip, whole-workgroup timing): a packed op occupies the SIMD for ~4.3 clocks against ~2.4
// for a plain one (~4.1 when a source is an SGPR, as the butterflies' constant twiddles are), so a butterfly on (re, im)
// pairs costs about 10-30 % less VALU time than the scalar form -- not half.  hipcc does not fold the (im, -re) swizzles
// of a complex multiply into op_sel on its own (tools/ubench/dif.hip: 218 extra moves per 32-point DIF), hence the asm.
// (-i) * (x - y) = (x.im - y.im, y.re - x.re)
__device__ __forceinline__ v2 rotSub(v2 x, v2 y)
{
    v2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// d * (k.x - i k.y), k wave-uniform (an SGPR pair): (d.re c + d.im s, d.im c - d.re s)
__device__ __forceinline__ v2 cmulConjK(v2 d, v2 k)
{
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(d), "s"(k));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(d), "s"(k), "v"(t));
    return r;
}
// c * w for a per-lane w = (w.re, w.im): (c.re w.re - c.im w.im, c.re w.im + c.im w.re)
__device__ __forceinline__ v2 cmul(v2 c, v2 w)
{
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(c), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(c), "v"(w), "v"(t));
    return r;
}

// In-register radix-2 DIF over LEN packed complex elements starting at BASE; result in bit-reversed order.
template <int R, int LEN, int BASE>
__device__ __forceinline__ void difPacked(v2 (&c)[R])
{
    constexpr int H = LEN / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const int a = BASE + i, b = BASE + i + H;
        const v2 x = c[a], y = c[b];
        c[a] = x + y;
        const int j = i * (32 / LEN);
        if (j == 0) c[b] = x - y;
        else if (j == 8) c[b] = rotSub(x, y);
        else c[b] = cmulConjK(x - y, v2{cos32(j), sin32(j)});
    }
    if constexpr (LEN > 2) {
        difPacked<R, H, BASE>(c);
        difPacked<R, H, BASE + H>(c);
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

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is fence(all address spaces) + s_barrier: the fence makes
// every wave wait for its outstanding GLOBAL loads too (s_waitcnt vmcnt(0)) -- the map tables and twiddles that are prefetched
// across a barrier on purpose.  With the fence restricted to the local address space only LDS traffic is waited for.
// The explicit s_waitcnt is NOT redundant: the fence makes the compiler wait for the LDS operations IT knows of, and the exchange-1
// stores (ldsWrite64 / ldsWrite128 below) are inline assembly -- invisible to its counters.  Until round 6 the barrier behind exchange 1's
// SECOND store round had no wait in front of it (the first one had, by the luck of a compiler-visible table store beside it): a wave
// could pass the barrier with its ds_write_b128 still in flight and a reader of another wave could be served first.  With one launch
// on the chip that practically never happened (0 of 20 000 fuzz cases); with several launches in flight it did, in 1-2 of 1 000 launches:
// one workgroup's transform with a stale value in it -- a faint broadband error in one frame (tools/ka_overlap_stress.py).
__device__ __forceinline__ void ldsBarrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS reads the load / store optimiser cannot pair up.  Left to itself hipcc merges two ds_read_b64 of one base into ds_read2_b64 (and
// ds_read2st64_b64), which the LDS serves at HALF the rate of two single reads (MI355X_MICROARCH.md, LDS table: 8 cycles per
// wave-instruction against 2 + 2).  ldsRead64<OFF>(addr): one ds_read_b64 at LDS byte address addr + OFF; the values are only valid
// behind ldsReadsDone() on them (the compiler's wait-count bookkeeping does not see inline assembly).
__device__ __forceinline__ uint32_t ldsAddress(const void *p)
{
    return uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)p));
}
template <int OFF>
__device__ __forceinline__ v2 ldsRead64(uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    v2 r;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ float ldsRead32(uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    float r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF>
__device__ __forceinline__ void ldsWrite64(uint32_t addr, v2 v)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
typedef float v4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void ldsWrite128(uint32_t addr, v2 lo, v2 hi)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    const v4 v = v4{lo.x, lo.y, hi.x, hi.y};
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ldsReadsDone(v2 (&t)[16])
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]),
                   "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]) : : "memory");
}
template <int N, int STRIDE, int I = 0>
__device__ __forceinline__ void ldsReadRun64(v2 (&t)[16], uint32_t addr)
{
    if constexpr (I < N) { t[I] = ldsRead64<I * STRIDE>(addr); ldsReadRun64<N, STRIDE, I + 1>(t, addr); }
}

// Four (re, im) pairs <- the same registers of lane L ^ 32, on the vector ALUs: v_permlane32_swap_b32 vdst, src exchanges vdst's upper
// half-wave with src's lower one; swap(x, y) then swap(y, x) leaves x's halves exchanged in y and y's in x.  The leading s_nop covers the
// two wait states a swap needs behind a vector write of its operands (inline assembly is invisible to the hazard recogniser); a pair's
// second swap follows its first at a distance of four.  (The builtin form -- r1 = __builtin_amdgcn_permlane32_swap(x, y, ..); r2 =
// ..swap(r1[1], r1[0], ..) -- is miscompiled by ROCm 7.2's hipcc, which copies y over x in front of the first swap:
// tools/ubench/permswap.hip.)
__device__ __forceinline__ void halfWaveExchange4(v2 &a, v2 &b, v2 &c, v2 &d)
{
    float x0 = a.x, y0 = a.y, x1 = b.x, y1 = b.y, x2 = c.x, y2 = c.y, x3 = d.x, y3 = d.y;
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                 "v_permlane32_swap_b32 %1, %0\n\tv_permlane32_swap_b32 %3, %2\n\tv_permlane32_swap_b32 %5, %4\n\tv_permlane32_swap_b32 %7, %6"
                 : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2), "+v"(x3), "+v"(y3));
    a = v2{y0, x0}; b = v2{y1, x1}; c = v2{y2, x2}; d = v2{y3, x3};           // (the pairs come out crossed)
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
// Global load at a wave-uniform base + a 32-bit per-lane byte offset: the form that selects `global_load … v_off, s[base]`
// (one offset VGPR; a 64-bit element index makes hipcc build, keep and spill a 64-bit address pair per load).
template <typename T>
__device__ __forceinline__ T ldg(const T *base, uint32_t byteOff)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byteOff);
}

// The same with the wave-uniform part pinned to a scalar register pair: `global_load v, v_lane, s[base]` for every element of an unrolled
// batch whose compile-time offsets do not fit the instruction's 13-bit immediate.  Left alone, hipcc keeps ONE 64-bit vector address and
// adds each offset to it with a v_add_co / v_addc pair per load (40 vector instructions for the 32 sample loads of a channel workgroup).
#ifndef SGZ_LOADS
#define SGZ_LOADS 1
#endif
template <typename T>
__device__ __forceinline__ T ldgPinned(const void *uniformBase, uint32_t uniformBytes, uint32_t laneBytes)
{
#if SGZ_LOADS == 0
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(uniformBase) + uniformBytes + laneBytes);
#elif SGZ_LOADS == 2
    const char *b = reinterpret_cast<const char *>(uniformBase) + uniformBytes;
    asm("" : "+s"(b));
    return *reinterpret_cast<const T *>(b + laneBytes);
#else
    // (the pointer keeps its GLOBAL address space through the asm: a generic pointer comes out as flat_load)
    typedef const char __attribute__((address_space(1))) *GlobalBytes;
    typedef const T __attribute__((address_space(1))) *GlobalT;
    GlobalBytes b = (GlobalBytes)(uniformBase) + uniformBytes;
    asm("" : "+s"(b));
    return *(GlobalT)(b + laneBytes);
#endif
}

// ODD (the odd half of a 2 R^3-point frame): every product carries the extra factor U = W_{2N}^{x}; the table has one
// more row, B_0 = U, and B_a = W^{x 4a} U.
template <int LR, bool ODD = false>
struct TwFactors {
    static constexpr int R = 1 << LR;
    static constexpr int NB = R / 4 - 1 + (ODD ? 1 : 0);
    float2 a[3];
    float2 b[NB];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int voff, int rowBytes)
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = bufLoad2(rs, voff, i * rowBytes);
#pragma unroll
        for (int i = 0; i < NB; ++i) b[i] = bufLoad2(rs, voff, (3 + i) * rowBytes);
    }
    // plain global loads (dwordx2): measured ~1.8x the throughput of raw buffer loads on gfx950 (tools/ubench/stream.hip)
    __device__ __forceinline__ void load(const float2 *tab, int idx, int rowElems)
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = ldg(tab, uint32_t(idx + i * rowElems) * 8u);
#pragma unroll
        for (int i = 0; i < NB; ++i) b[i] = ldg(tab, uint32_t(idx + (3 + i) * rowElems) * 8u);
    }
    // packed form: c[i] = (re, im)
    __device__ __forceinline__ void apply(v2 (&c)[R]) const
    {
#pragma unroll
        for (int q = ODD ? 0 : 1; q < R; ++q) {
            const int qa = q >> 2, qb = q & 3;
            const int ib = ODD ? qa : qa - 1;                          // row of B_qa
            v2 w;
            if (!ODD && qa == 0) w = v2{a[qb - 1].x, a[qb - 1].y};
            else if (qb == 0) w = v2{b[ib].x, b[ib].y};
            else w = cmul(v2{b[ib].x, b[ib].y}, v2{a[qb - 1].x, a[qb - 1].y});
            const int i = brev(q, LR);
            c[i] = cmul(c[i], w);
        }
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

}  // namespace sgz
