// spectrum_real16.hip -- K_A, channel-split form at N = 32768 with 1024-thread workgroups: one workgroup per (frame, pair, channel),
// SIXTEEN complex values per thread instead of the thirty-two of spectrum_real.hip.  gfx950 only.
//
// Why a second body.  The 512-thread form needs 128 VGPRs (32 values + temporaries), so a CU holds two workgroups = sixteen waves,
// two per SIMD; a workgroup that is alone on its CU -- the partial last dispatch generation of a 348-frame launch -- leaves the
// vector ALUs idle behind its own dependent chains (14.5 us alone against 7.9 us per workgroup when two share the CU; NOTES.md).
// Sixteen values per thread fit 64 VGPRs: a workgroup is sixteen waves, two workgroups are thirty-two (eight per SIMD, the
// machine's limit), and a lone workgroup still puts four waves on every SIMD with half the chain each.
//
// The transform (same real-input scheme: M = N/2 = 16384 complex points z[n] = x[2n] + i x[2n+1], recombined afterwards;
// TransformDSP.inl:234-484 prepare, :487-502 FFT, :854-986 split + map are what it replaces).  M = 16 x 16 x (4 x 16);
// tools/emulate_real16.py walks these index formulas on the CPU against numpy's FFT:
//   pass 1   thread c = tid owns z[c + 1024 j], j < 16: radix 16 in registers -> q1, times W_M^{c q1}
//   exch 1   workgroup-wide through LDS in two rounds (q1 < 8, then q1 >= 8: 64 KB each): wave q1 receives its 1024 values,
//            lane = c_lo (6 bits), register = c_hi (4 bits), c = c_lo + 64 c_hi.  From here on a wave works alone.
//   pass 2   radix 16 over c_hi -> q2, times W_1024^{c_lo q2} (LDS table)
//   exch 2   wave-local transpose through LDS (planar, rows of 64 + 4 floats; reads are ds_read_b128): lane = 4 q2 + l,
//            register r, c_lo = 16 l + r
//   pass 3   radix 4 ACROSS the four lanes of a quad (two v_fmac_f32_dpp steps per float: own + sigma x partner; lane 3 turns
//            its value by i in between) -> m = brev2(l), times sign_l W_64^{r m} (LDS table, the signs of the quad stage folded in),
//            then radix 16 over r in registers -> s.   Z[k] with k = q1 + 16 q2 + 256 m + 1024 s sits in register brev4(s).
//   mirror   Z[M - k] is in wave 16 - q1, lane 63 - lane, register brev4(15 - s): every wave parks its registers s >= 8 in LDS
//            (its own exchange-2 tile), one barrier, and every thread evaluates the bin pairs (k, M - k) of its registers s < 8
//            exactly as spectrum_real.hip does -- both magnitudes out of one packed multiply and one packed multiply-add.
//   map      chunk_map.hpp's chunk-scan map on the same tables and the same LDS layout of |X|; two adjacent lanes share a row of 32
//            magnitudes, the odd lane takes over the even lane's running maximum (ChunkMap16 below).
// Everything a (frame, pair, channel) leaves behind -- mapped pixels, ny, nyBest, low -- is what spectrum_real.hip leaves: the late-pixel
// kernels and K_B do not know which form ran.
#include <algorithm>

#include "real_common.hpp"

namespace sgz {

namespace {

constexpr int kT = 1024, kM = 16384, kN = 32768;
constexpr int kXFloats = realXFloats(kM);                     // a side's |X| array (17426 floats); the exchange areas live in the same floats
constexpr int kRow = 68, kTile = 16 * kRow;                   // exchange 2: a wave's planar tile, 16 rows of 64 + 4 floats
constexpr int kTab = (kXFloats + 3) & ~3;                     // pass-2 table [16][64] float2, pass-3 table [4][16] float2 (16-byte aligned)
constexpr int kTabFloats = (16 * 64 + 4 * 16) * 2;
constexpr int kScratch = kTab + kTabFloats;                   // column 0's 64 complex values
constexpr int kLdsFloats = kScratch + 128;
static_assert(16 * kTile <= kTab, "the exchange-2 tiles end below the twiddle tables");

template <int CTRL>
__device__ __forceinline__ float dppMove(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// own + sigma x (the value of the lane two / one further in the quad), on eight floats in place.  The leading s_nop covers the two
// wait states a DPP read needs behind a vector write of the same register (inline assembly is invisible to the hazard recogniser);
// inside the block every instruction reads a register no earlier instruction of the block wrote.
#define SGZ_QUAD_STEP(PERM)                                                                                             \
    asm volatile("s_nop 1\n\t"                                                                                        \
                 "v_fmac_f32_dpp %0, %0, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_fmac_f32_dpp %1, %1, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_fmac_f32_dpp %2, %2, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_fmac_f32_dpp %3, %3, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_fmac_f32_dpp %4, %4, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_fmac_f32_dpp %5, %5, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_fmac_f32_dpp %6, %6, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                         \
                 "v_fmac_f32_dpp %7, %7, %8 quad_perm:" PERM " row_mask:0xf bank_mask:0xf"                             \
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(sigma))
__device__ __forceinline__ void quadStepFar(float &f0, float &f1, float &f2, float &f3, float &f4, float &f5, float &f6, float &f7, float sigma)
{
    SGZ_QUAD_STEP("[2,3,0,1]");
}
__device__ __forceinline__ void quadStepNear(float &f0, float &f1, float &f2, float &f3, float &f4, float &f5, float &f6, float &f7, float sigma)
{
    SGZ_QUAD_STEP("[1,0,3,2]");
}
#undef SGZ_QUAD_STEP

// The chunk-scan map of chunk_map.hpp for 1024 threads: lanes 2 r and 2 r + 1 share row r (32 magnitudes), sixteen each.  Both run
// the segmented running maximum over their half; the odd lane then takes the even lane's last running value as the carry into
// its own open run (unless the row's element 15 closed a tile): into the first tile it closes (an LDS float maximum on that slot,
// behind this lane's own store to it) or, with no tile end in its half, into the row's running value CE[r].
struct ChunkMap16 {
    uint2 cr;
    uint32_t slotBase, endBits, rowEnds;

    __device__ __forceinline__ void prefetch(const ChunkTables &tb, int tid)
    {
        const int row = tid >> 1;
        rowEnds = tb.ends[row];
        const uint32_t base = tb.reBase[row];
        endBits = (tid & 1) ? rowEnds >> 16 : rowEnds & 0xFFFFu;
        slotBase = base + ((tid & 1) ? uint32_t(__builtin_popcount(rowEnds & 0xFFFFu)) : 0u);
        cr = tid < tb.P ? tb.crec[tid] : uint2{0u, kChunkOff};
    }

    template <typename Index>
    __device__ __forceinline__ void run(const ChunkTables &tb, const Index at, const float *lds, float *re, float *ce, float invSize, int tid)
    {
        using Wide = ChunkMap<512>;
        float4 wq[3];
        const bool interp = Wide::interpolated(cr);
        const bool anyInterp = __builtin_amdgcn_ballot_w64(interp) != 0ull;
        if (anyInterp) {
            const float4 *wp = reinterpret_cast<const float4 *>(tb.weights12 + size_t(interp ? cr.x : 0u) * kTapFloats);
            wq[0] = wp[0]; wq[1] = wp[1]; wq[2] = wp[2];
        }
        float v[16];
        {
            const float2 *src = reinterpret_cast<const float2 *>(lds + chunkPos(32 * (tid >> 1)) + 16 * (tid & 1));
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float2 t = src[j]; v[2 * j] = t.x; v[2 * j + 1] = t.y; }
        }
        v[0] = __builtin_fabsf(v[0]);                                   // csf[0] (left side, row 0) is a signed real; every other entry is a magnitude
        const uint32_t reAddr = uint32_t(uintptr_t((__attribute__((address_space(3))) const void *)re));
        const uint32_t firstSlot = (slotBase << 2) + reAddr;
        uint32_t slotAddr = firstSlot;
        uint64_t prev = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t e = __builtin_amdgcn_uicmp(endBits & (1u << j), 0u, 33 /* ICMP_NE */);
            if (j > 0)
                asm volatile("s_andn2_b64 exec, exec, %2\n\tv_max_f32 %0, %0, %1\n\ts_mov_b64 exec, -1"
                             : "+v"(v[j]) : "v"(v[j - 1]), "s"(prev) : "scc");
            if (e != 0)                                                 // wave-uniform
                asm volatile("s_mov_b64 exec, %2\n\tds_write_b32 %0, %1\n\tv_add_u32 %0, 4, %0\n\ts_mov_b64 exec, -1"
                             : "+v"(slotAddr) : "v"(v[j]), "s"(e) : "memory");
            prev = e;
        }
        // ---- the even lane's open run continues in the odd lane
        {
            const float carry = dppMove<0xB1>(v[15]);                   // quad_perm [1,0,3,2]: the neighbour's last running value
            const bool takes = (tid & 1) && !(rowEnds & 0x8000u);
            float fin = v[15];
            if (takes && endBits != 0u) asm volatile("ds_max_f32 %0, %1" : : "v"(firstSlot), "v"(carry) : "memory");
            if (takes && endBits == 0u) fin = __builtin_fmaxf(fin, carry);
            if (tid & 1) ce[tid >> 1] = fin;
        }
        // ---- this thread's interpolated pixel
        if (anyInterp) {
            const float acc = Wide::taps(lds, interp ? (cr.y & 0xFFFFu) : 0u, wq);
            if (interp && tb.out) tb.out[tid] = finishPixel<5>(invSize * acc);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the tile stores above are invisible to the compiler's counters
        ldsBarrier();
        Wide::resolve(tb, at, lds, re, ce, invSize, cr, tid);
        for (int base = kT; base < tb.P; base += kT) {                  // more than 1024 pixels per side
            const int x = base + tid;
            if (x < tb.P) {
                const uint2 c = tb.crec[x];
                if (Wide::interpolated(c)) {
                    const float4 *wp = reinterpret_cast<const float4 *>(tb.weights12 + size_t(c.x) * kTapFloats);
                    const float4 q[3] = {wp[0], wp[1], wp[2]};
                    const float acc = Wide::taps(lds, c.y & 0xFFFFu, q);
                    if (tb.out) tb.out[x] = finishPixel<5>(invSize * acc);
                }
                Wide::resolve(tb, at, lds, re, ce, invSize, c, x);
            }
        }
    }
};

__device__ __forceinline__ ChunkTables tablesOf(const RealParams &prm, int side, long task, long self)
{
    constexpr int rows = kM / 32;
    return ChunkTables{prm.chunkEnds + side * rows, prm.chunkReBase + side * rows,
                       reinterpret_cast<const uint2 *>(prm.chunkRec) + size_t(side) * prm.P, prm.recs + size_t(side) * prm.P, prm.weights12,
                       prm.mapped ? prm.mapped + (size_t(task) * 2 + side) * prm.P : nullptr, prm.nyBest + size_t(self) * 64, int(prm.fixFrom[side]),
                       int(prm.P), side != 0, nullptr};
}

// behind the barrier that completes a side's magnitudes in LDS (the counterpart of realMapSettle in spectrum_real.hip)
__device__ __forceinline__ void mapSettle16(const RealParams &prm, float *lds, const int tid, const int side, const long task, const long self,
                                            const ChunkTables &tb, ChunkMap16 &mapper, float *re, float *ce)
{
    const ChannelIndex at{kN, side ? kM : 0};
    if (prm.binsOut) {                                                      // test hook: this side's half of csf, csf order (csf[N/2] is the late kernel's)
        float *dst = prm.binsOut + size_t(task) * (kN + 1) + (side ? kM : 0);
        for (int i = tid; i <= kM; i += kT)
            if (i != (side ? 0 : kM)) dst[i] = lds[chunkPos(i)];
    }
    if ((prm.lowCount[0] + prm.lowCount[1]) && tid < kLowBins) prm.low[size_t(self) * kLowBins + tid] = lds[at(side ? kN - tid : tid)];
    mapper.run(tb, at, lds, re, ce, prm.invSize, tid);
}

}  // namespace

template <bool WCOS, int MIX>                                 // MIX: 0 Separate, 2 MidSide (mid and side signals)
__global__ void __launch_bounds__(1024, 8) stftReal16Kernel(const RealParams prm)
{
    constexpr bool mixed = MIX == 2;
    constexpr int PADSTRIDE = chunkPos(1024);                   // padded distance between entries k and k + 1024
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    const UnitId u = unitOf<false>(prm);                    // (frame, pair, channel); real_common.hpp
    const int side = u.side, pair = u.pair;
    const long task = u.task, frame = u.frame, self = u.self;

    const uint32_t maxSlots = prm.chunkSlots[0] > prm.chunkSlots[1] ? prm.chunkSlots[0] : prm.chunkSlots[1];
    const ChunkTables tb = tablesOf(prm, side, task, self);
    float *re = lds + kXFloats, *ce = re + maxSlots + 1;
    ChunkMap16 mapper;

    // every table value needed before the first barrier is requested behind the samples and in front of the first wait (see spectrum_real.hip)
    constexpr bool FRONT = !mixed;                             // (MidSide holds a second channel's samples on top: it would spill)
    [[maybe_unused]] float4 phase, tabPiece;
    [[maybe_unused]] float2 twA, twB;
    if constexpr (!FRONT) {
        if (tid < kTabFloats / 4) reinterpret_cast<float4 *>(lds + kTab)[tid] = prm.tw16[tid];
    }
    v2 c[16];
    {
        // ------------------------------------------------------------ load + window: z[n] = (x[2n] w[2n], x[2n+1] w[2n+1]), n = tid + 1024 j
        const int firstCh = mixed ? 0 : side;
        const float *X = prm.planar + size_t(2 * pair + firstCh) * prm.chStride + size_t(frame) * prm.hop;
        const uint32_t lane8 = uint32_t(tid) * 8u;
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = ldgPinned<v2>(X, 8192u * uint32_t(j), lane8);
        if constexpr (FRONT) {
            if (WCOS) phase = ldg(prm.winPhase, uint32_t(tid) * 16u);     // p1 x (cos even, cos odd, sin even, sin odd)
            twA = ldg(prm.tw1, uint32_t(tid) * 8u);
            twB = ldg(prm.tw1 + 3 * 1024, uint32_t(tid) * 8u);
            if (tid < kTabFloats / 4) tabPiece = prm.tw16[tid];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (mixed) {
            // (l +- r) w 0.5 (prepareTransform, TransformDSP.inl:92-135)
            const float *Y = X + prm.chStride;
            const float sgn = side == 1 ? -1.f : 1.f;
#pragma unroll
            for (int b0 = 0; b0 < 16; b0 += 8) {
                float2 y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = ldg(reinterpret_cast<const float2 *>(Y) + 1024 * (b0 + i), lane8);
#pragma unroll
                for (int i = 0; i < 8; ++i) c[b0 + i] = v2{c[b0 + i].x + sgn * y[i].x, c[b0 + i].y + sgn * y[i].y};
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (WCOS) {
            // w[n] = p0 + p1 cos(theta_n), theta = phi(column, e) + 2 pi j / 16: the column's phase from the plan's table, the step to
            // the next pair a rotation by a compile-time angle; j + 8 is half a turn further: w[j] = p0 + t, w[j + 8] = p0 - t
            const v2 p0 = v2{prm.winP0, prm.winP0};
            const float4 ph = FRONT ? phase : ldg(prm.winPhase, uint32_t(tid) * 16u);
            const v2 pc = v2{ph.x, ph.y}, ps = v2{ph.z, ph.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int a32 = 2 * j;                                  // angle in 32nds of a turn
                v2 t = pc;
                if (a32 != 0) {
                    const v2 k = v2{cos32(a32), sin32(a32)};
                    v2 m;
                    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m) : "v"(ps), "s"(k));                   // ps sin
                    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(pc), "s"(k), "v"(m));   // pc cos - ps sin
                }
                v2 wa, wb;
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(wa) : "v"(t), "s"(p0));
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0]" : "=v"(wb) : "v"(t), "s"(p0));
                c[j] = c[j] * wa;
                c[j + 8] = c[j + 8] * wb;
            }
        } else {
#pragma unroll
            for (int b0 = 0; b0 < 16; b0 += 8) {
                float2 w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = ldg(reinterpret_cast<const float2 *>(prm.window) + 1024 * (b0 + i), lane8);
#pragma unroll
                for (int i = 0; i < 8; ++i) c[b0 + i] = v2{c[b0 + i].x * w[i].x, c[b0 + i].y * w[i].y};
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (mixed) {
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = v2{c[i].x * 0.5f, c[i].y * 0.5f};
    }
    // -------------------------------------------------------------------------- pass 1: radix 16, times W_M^{c q1}
    ditPacked<4, 0>(c);
    {
        // W^c and W^{4c} from the table, the other rows as their powers (as spectrum_real.hip)
        auto sq = [](float2 w) { return float2{w.x * w.x - w.y * w.y, 2.f * w.x * w.y}; };
        auto mul = [](float2 p, float2 q) { return float2{p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x}; };
        float2 a[3], b[3];
        a[0] = FRONT ? twA : ldg(prm.tw1, uint32_t(tid) * 8u);
        b[0] = FRONT ? twB : ldg(prm.tw1 + 3 * 1024, uint32_t(tid) * 8u);
        a[1] = sq(a[0]); a[2] = mul(a[1], a[0]);
        b[1] = sq(b[0]); b[2] = mul(b[1], b[0]);
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const int qa = q >> 2, qb = q & 3;
            v2 w;
            if (qa == 0) w = v2{a[qb - 1].x, a[qb - 1].y};
            else if (qb == 0) w = v2{b[qa - 1].x, b[qa - 1].y};
            else w = cmul(v2{b[qa - 1].x, b[qa - 1].y}, v2{a[qb - 1].x, a[qb - 1].y});
            c[brev(q, 4)] = cmul(c[brev(q, 4)], w);
        }
    }
    // -------------------------------------------------------------------------- exchange 1: two rounds of 8 x 1024 complex values
    v2 d[16];
    {
        v2 *lds2 = reinterpret_cast<v2 *>(lds);
        const int rd = (wave & 7) * 1024 + lane;
        if constexpr (FRONT) { if (tid < kTabFloats / 4) reinterpret_cast<float4 *>(lds + kTab)[tid] = tabPiece; }      // the pass-2 / pass-3 twiddle tables -> LDS
#pragma unroll
        for (int q = 0; q < 8; ++q) lds2[q * 1024 + tid] = c[brev(q, 4)];
        ldsBarrier();
        if (wave < 8) {
#pragma unroll
            for (int h = 0; h < 16; ++h) d[h] = lds2[rd + 64 * h];
        }
        ldsBarrier();
#pragma unroll
        for (int q = 8; q < 16; ++q) lds2[(q - 8) * 1024 + tid] = c[brev(q, 4)];
        ldsBarrier();
        if (wave >= 8) {
#pragma unroll
            for (int h = 0; h < 16; ++h) d[h] = lds2[rd + 64 * h];
        }
        ldsBarrier();                                                    // every wave has read: the tiles may overwrite the area
    }
    // -------------------------------------------------------------------------- pass 2 (wave q1 = wave, lane = c_lo): radix 16 over c_hi
    ditPacked<4, 0>(d);
    {
        const v2 *tab = reinterpret_cast<const v2 *>(lds + kTab) + lane;
#pragma unroll
        for (int q = 1; q < 16; ++q) d[brev(q, 4)] = cmul(d[brev(q, 4)], tab[q * 64]);
    }
    // -------------------------------------------------------------------------- exchange 2: wave-local, planar
    v2 e[16];
    {
        float *tile = lds + wave * kTile;
        const float4 *rp = reinterpret_cast<const float4 *>(tile + (lane >> 2) * kRow + (lane & 3) * 16);
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[q * kRow + lane] = d[brev(q, 4)].x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float4 t = rp[i]; e[4 * i].x = t.x; e[4 * i + 1].x = t.y; e[4 * i + 2].x = t.z; e[4 * i + 3].x = t.w; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[q * kRow + lane] = d[brev(q, 4)].y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float4 t = rp[i]; e[4 * i].y = t.x; e[4 * i + 1].y = t.y; e[4 * i + 2].y = t.z; e[4 * i + 3].y = t.w; }
    }
    // -------------------------------------------------------------------------- pass 3 (lane = 4 q2 + l): radix 4 across the quad, radix 16 over r
    const int q2 = lane >> 2, l4 = lane & 3;
    const int m4 = ((l4 & 1) << 1) | (l4 >> 1);                          // the quad stage's output index in this lane
    {
        const float sFar = (l4 & 2) ? -1.f : 1.f;                        // (y0 + y2, y1 + y3, y2 - y0, y3 - y1)
        const float sNear = (l4 == 1 || l4 == 2) ? -1.f : 1.f;
        const bool turn = l4 == 3;
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            float f[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { f[2 * i] = e[g + i].x; f[2 * i + 1] = e[g + i].y; }
            quadStepFar(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], sFar);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                               // lane 3: times i
                const float xr = f[2 * i], xi = f[2 * i + 1];
                f[2 * i] = turn ? -xi : xr;
                f[2 * i + 1] = turn ? xr : xi;
            }
            quadStepNear(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], sNear);
#pragma unroll
            for (int i = 0; i < 4; ++i) e[g + i] = v2{f[2 * i], f[2 * i + 1]};
        }
        const v2 *tab = reinterpret_cast<const v2 *>(lds + kTab + 16 * 64 * 2) + l4 * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = cmul(e[r], tab[r]);
    }
    const int kb = wave + 16 * q2 + 256 * m4;                            // this thread's bins: k = kb + 1024 s, s in register brev4(s)
    const float2 wk = ldg(prm.twPost16, uint32_t(kb) * 8u);              // W_N^{kb}; a bin's is W_N^{kb} W_32^s
    ditPacked<4, 0>(e);
    // ---- column 0 (q1 = q2 = 0: lanes 0 .. 3 of wave 0, k = 256 (m + 4 s)) pairs bins inside the quad and holds DC / Nyquist: the quad
    // leaves its 64 values in scratch, lanes 0 .. 32 of wave 0 redo those bins and store them after the workgroup's own stores
    if (tid < 4) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            lds[kScratch + 2 * (m4 + 4 * s)] = e[brev(s, 4)].x;
            lds[kScratch + 2 * (m4 + 4 * s) + 1] = e[brev(s, 4)].y;
        }
        if (tid == 0) prm.ny[self] = 2.f * (e[0].x - e[0].y);            // this channel's Nyquist bin X[M] = Re Z[0] - Im Z[0] (the transform runs on x w / 2)
    }
    float fixA = 0.f, fixB = 0.f;
    if (tid <= 32) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (tid >= 1) {
            const int j = tid;                                              // k1 = 256 j and its mirror k2 = 256 (64 - j)  (j = 32: one bin)
            const v2 a = v2{lds[kScratch + 2 * j], lds[kScratch + 2 * j + 1]};
            const v2 b = v2{lds[kScratch + 2 * (64 - j)], lds[kScratch + 2 * (64 - j) + 1]};
            const float cs = cospif(float(j) * (1.0f / 64.0f)), sn = sinpif(float(j) * (1.0f / 64.0f));   // W_N^{256 j} = (cos, -sin)
            fixA = realBinMag(a, b, v2{cs, -sn});
            fixB = realBinMag(b, a, v2{-cs, -sn});
        } else {
            fixA = lds[kScratch] + lds[kScratch + 1];                    // csf[0] = X_c[0] / 2, signed (TransformDSP.inl:861-862; the 1/2 came in with the window)
        }
    }
    // ---- mirror: registers s >= 8 -> this wave's tile (its exchange-2 reads are behind it), one barrier, then Z[M - k] of the
    // registers s < 8 from wave 16 - q1
    {
        v2 *mz = reinterpret_cast<v2 *>(lds + wave * kTile) + lane;
#pragma unroll
        for (int s = 8; s < 16; ++s) mz[(s - 8) * 64] = e[brev(s, 4)];
    }
    ldsBarrier();
    float magA[8], magB[8];
    {
        const int pw = (16 - wave) & 15;
        const int pl = (wave == 0 && lane >= 4) ? 67 - lane : 63 - lane;  // q1 = 0: q2' = 16 - q2 (the quad q2 = 0 is column 0: redone above)
        const v2 *pz = reinterpret_cast<const v2 *>(lds + pw * kTile) + pl;
        v2 bb[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) bb[s] = pz[(7 - s) * 64];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const v2 a = e[brev(s, 4)], b = bb[s];
            const v2 w = s == 0 ? v2{wk.x, wk.y} : cmulConjK(v2{wk.x, wk.y}, v2{cos32(s), sin32(s)});
            // (the packed form of spectrum_real.hip's recombination: E = a + conj b, D = a - conj b, O = -i w D; real and imaginary
            // parts of 2 X[k], 2 X[M - k] pairwise)
            v2 E, D, t, O, re2, im2, sq;
            asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(E) : "v"(a), "v"(b));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(D) : "v"(a), "v"(b));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(w), "v"(D));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(O) : "v"(w), "v"(D), "v"(t));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(re2) : "v"(E), "v"(O));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(im2) : "v"(E), "v"(O));
            sq = re2 * re2;
            asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(sq) : "v"(im2), "v"(sq));
            magA[s] = __builtin_amdgcn_sqrtf(sq.x);
            magB[s] = __builtin_amdgcn_sqrtf(sq.y);
        }
    }
    // csf[N/2 - 1] *= 0.5 (quirk Q3, TransformDSP.inl:864): the left channel's bin M - 1 = the mirror of bin 1
    if (side == 0 && kb == 1) magB[0] *= 0.5f;
    mapper.prefetch(tb, tid);
    ldsBarrier();                                                        // the parked registers are dead: |X| may overwrite them
    {
        // left: bin k at position k; right: at position M - k (csf order is ascending in LDS on both sides)
        const int up = chunkPos(kb), down = chunkPos(kM - kb);
        int lowest = down - 7 * PADSTRIDE;
        asm volatile("" : "+v"(lowest));
        float *pu = lds + up, *pd = lds + lowest;
        if (side == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s) { pu[s * PADSTRIDE] = magA[s]; pd[(7 - s) * PADSTRIDE] = magB[s]; }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) { pd[(7 - s) * PADSTRIDE] = magA[s]; pu[s * PADSTRIDE] = magB[s]; }
        }
    }
    if (tid <= 32) {                                                        // column 0, after this wave's own stores (one wave's LDS operations execute in order)
        auto put = [&](int k, float v) { const int i = side ? kM - k : k; lds[chunkPos(i)] = v; };
        if (tid >= 1) {
            put(256 * tid, fixA);
            if (tid != 32) put(256 * (64 - tid), fixB);
        } else {
            put(0, fixA);
            put(kM, 0.f);                                                // csf[N/2] is settled late, 0 can never win meanwhile (strict >)
        }
    }
    // the two pad floats behind every row of 32 and the floats behind entry M: tap windows read over them with weight 0
    if (tid < 512) *reinterpret_cast<float2 *>(lds + chunkPos(32 * tid) + 32) = float2{0.f, 0.f};
    if (tid < 16) lds[chunkPos(kM) + 1 + tid] = 0.f;
    ldsBarrier();
    mapSettle16(prm, lds, tid, side, task, self, tb, mapper, re, ce);
}

// Test hook (sgz_stage_map_from_bins on a plan that runs this form): csf magnitudes [task][N + 1] come from HBM instead of the
// transform; the map behind them is mapSettle16, the code the transform kernel runs.
__global__ void __launch_bounds__(1024) realMapFromBins16Kernel(const RealParams prm)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const long unit = blockIdx.x;
    const int side = int(unit & 1);
    const long task = unit >> 1;
    const long self = (task << 1) | side;
    const uint32_t maxSlots = prm.chunkSlots[0] > prm.chunkSlots[1] ? prm.chunkSlots[0] : prm.chunkSlots[1];
    const ChunkTables tb = tablesOf(prm, side, task, self);
    float *re = lds + kXFloats, *ce = re + maxSlots + 1;
    ChunkMap16 mapper;
    mapper.prefetch(tb, tid);
    const float *src = prm.binsIn + size_t(task) * (kN + 1) + (side ? kM : 0);
    for (int i = tid; i <= kM; i += kT) lds[chunkPos(i)] = (i == (side ? 0 : kM)) ? 0.f : src[i];
    if (tid < 512) *reinterpret_cast<float2 *>(lds + chunkPos(32 * tid) + 32) = float2{0.f, 0.f};
    if (tid < 16) lds[chunkPos(kM) + 1 + tid] = 0.f;
    if (tid == 0) prm.ny[self] = src[side ? 0 : kM];
    __syncthreads();
    mapSettle16(prm, lds, tid, side, task, self, tb, mapper, re, ce);
}

hipError_t launchStftReal16(const RealParams &prm, hipStream_t stream)
{
    const long units = prm.frames * long(prm.C) * 2;
    if (units <= 0) return hipSuccess;
    const uint32_t maxSlots = std::max(prm.chunkSlots[0], prm.chunkSlots[1]);
    const size_t ldsBytes = real16LdsBytes(maxSlots);
    static_assert(size_t(kLdsFloats) * 4 == real16LdsBytes(0), "real_common.hpp states this kernel's LDS layout");
    if (ldsBytes > 80 * 1024) return hipErrorInvalidValue;
    static size_t granted[5][64] = {};
    auto go = [&](auto kern, int slot) -> hipError_t {
        if (hipError_t e = grantLds(reinterpret_cast<const void *>(kern), ldsBytes, granted[slot]); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(unsigned(units)), dim3(1024), ldsBytes, stream, prm);
        return hipGetLastError();
    };
    hipError_t e;
    if (prm.binsIn) {
        e = go(&realMapFromBins16Kernel, 4);
        return e != hipSuccess ? e : launchRealLate(prm, kN, stream);
    }
    const bool wcos = prm.winPhase != nullptr;
    if (prm.mode == SGZ_CH_MIDSIDE) e = wcos ? go(&stftReal16Kernel<true, 2>, 0) : go(&stftReal16Kernel<false, 2>, 1);
    else e = wcos ? go(&stftReal16Kernel<true, 0>, 2) : go(&stftReal16Kernel<false, 0>, 3);
    if (e != hipSuccess) return e;
    // the pixels that need both channels (skipped when the caller's next kernel overlays them itself: prm.lateInNext)
    if (!prm.lateInNext) return launchRealLate(prm, kN, stream);
    return hipSuccess;
}

}  // namespace sgz
