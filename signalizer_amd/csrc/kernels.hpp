// kernels.hpp -- launch interfaces between the C-ABI glue (api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.hpp"
#include "late_fix.hpp"

namespace sgz {

struct StftParams {
    const float *planar;      // device, channel c at planar + c*chStride
    size_t chStride;
    long frames;
    uint32_t hop, W, P, C;
    uint32_t sides, mode;
    const float *window;      // [N]
    // fused kernel, W == N, Hann / Hamming periodic: w[t + T j] = p0 + p1 cos(phi_t + 2 pi j / R) is evaluated from winPhase[t] = (cos, sin)
    // of phi_t = 2 pi t / N instead of fetched -- the window is a third of the kernel's L2 -> L1 traffic.  Null: fetch it.
    const float2 *winPhase; float winP0, winP1;
    const float2 *tw1;        // [R][T]
    const float2 *tw2;        // [R][R]
    const float2 *tw1odd;     // halves path (N = 2 R^3): pass-1 twiddles of the odd half, W_N^{t (2q+1)} factorised
    // Complex and the mono modes leave a few csf entries complex (complex_dc.hpp): the pixels that reach one (plan.cpp dcPixels)
    // are redone with the complex values after the magnitude-only mapping
    const uint32_t *dcPixels;
    uint32_t nDcPixels;
    float2 *dcOut;            // generic / halves path: those entries, [task][kSpecBins] (the fused kernel keeps them in LDS), or null
    float2 *zOut;             // stftComplexKernel: the raw transform, per task re[N] then im[N] (Phase mode's split kernel reads it)
    uint32_t binsSplit;       // halves path: binsOut holds [even bins 0..N/2 | odd bins] per task (launchMapSides) instead of csf order
    long taskBase;            // halves path: first (frame, pair) task of this launch (outputs are indexed from 0)
    const PixelRec *recs;     // [sides][P]
    const float *weights;
    const float *weights11;   // fused kernel: kMaxTaps + 1 floats per record, zero where the tap window steps over an LDS pad slot
    const MaxItem *items;     // balanced arg-max work list (null: serial per-pixel scan)
    uint32_t nItems;
    uint32_t nItemsLeft;      // items [0, nItemsLeft) belong to left-side records (ascending k), the rest to the right side
    float invSize;
    float *mapped;            // [frames][C][sides][P] or null
    float *binsOut;           // [frames][C][N+1] or null (test hook)
    const float *binsIn;      // test hook: skip the FFT, map from these bins
    uint32_t ablate;          // debug: bit0 skip DIFs, bit1 skip exchange 1, bit2 skip exchange 2, bit3 skip mirror/M, bit4 skip map, bit5 skip twiddles
    unsigned long long *phaseClock;   // debug hook: workgroup 0 stores s_memtime at phase boundaries (16 slots) or null
    uint32_t roundSize;       // workgroups that run concurrently (number of CUs), for the frame -> workgroup order
};
// channel-split K_A (spectrum_real.hip): one workgroup per (frame, pair, channel), Separate mode, N = 32768 / 65536, W == N
struct RealParams {
    const float *planar;      // device, channel c at planar + c*chStride (8-byte aligned rows, even hop: the samples are fetched in pairs)
    size_t chStride;
    long frames;
    uint32_t hop, C, P;
    uint32_t mode;            // SGZ_CH_SEPARATE (two channel workgroups per task), or Left / Right / Merge / Side (one workgroup: the mixed signal)
    const float *window;      // [N]
    // Hann / Hamming, periodic: w[n] = p0 + p1 cos(2 pi n / N) is evaluated in the kernel instead of fetched (a third of the kernel's
    // L2 -> L1 traffic is the window, the same 4 N bytes for every workgroup): winPhase[c] = (cos, sin) of 2 pi (2c) / N and of
    // 2 pi (2c + 1) / N for the 1024 columns; the step to a column's next sample pair is a compile-time rotation.  Null: fetch.
    const float4 *winPhase; float winP0, winP1;
    const float2 *tw1;        // [3 + R1/4 - 1][1024]  pass-1 twiddles W_{N/2}^{c q}
    const float2 *tw2;        // [10][32]  (the whole-frame kernels' pass-2 table: N = 16384 builds the other rows as products)
    const float4 *tw2Full;    // N >= 32768: W_1024^{c q} as [c < 32][34] float2 (q < 32, two pad entries), copied to LDS by every workgroup: a thread's 32 factors are 16 ds_read_b128
    const float2 *twPost;     // [R1 * 32]  W_N^{kc}
    const float4 *tw16;       // N = 32768, 1024-thread form (spectrum_real16.hip): [16][64] + [4][16] float2, staged in LDS; null: the 512-thread form
    const float2 *twPost16;   // [1024] W_N^{kb}
    const PixelRec *recs; const float *weights;
    // the chunk-scan pixel map (chunk_map.hpp; plan.cpp buildChunkMap): per side, [T] end bits, [T] slot bases, [P] records
    const uint32_t *chunkEnds; const uint32_t *chunkReBase; const uint32_t *chunkRec; const float *weights12;
    uint32_t chunkSlots[2];
    float invSize;
    float *mapped;            // [frames][C][2][P] or null
    float *binsOut;           // test hook: [frames][C][N+1] csf magnitudes, or null
    const float *binsIn;      // test hook (pairs only): skip the transform, map from these csf magnitudes [frames][C][N+1]
    // csf[N/2] = |X_L[M] + i X_R[M]| / 2 needs both channels.  Nobody waits for it: a workgroup maps with 0 there and leaves its Nyquist bin
    // (ny, [unit]) and the winning squares of the pixels whose arg-max run ends on csf[N/2] (nyBest, [unit][64]) in HBM; realLateKernel,
    // launched behind the channel workgroups, settles those pixels for both sides.
    float *ny; float *nyBest;
    uint32_t lateInNext;      // 1: the caller's next kernel applies the late pixels itself (no realLateKernel launch)
    uint32_t fixFrom[2];      // per side: pixels [fixFrom, P) have runs that end on csf[N/2]
    // Interpolated pixels whose tap window reaches over bin 0 into the OTHER channel's half of csf (the window's periodic indexing:
    // ..., csf[N-1], csf[N], csf[0], csf[1], ...).  Settled like csf[N/2]: both channels publish their kLowBins lowest csf entries
    // (low, [unit][kLowBins]: left csf[j], right csf[N - j]), map with those pixels switched off (recs = the plan's channel-split copy),
    // and the later workgroup evaluates the listed pixels of both sides from the published entries (recsFull / weights).
    float *low;
    const PixelRec *recsFull; // the unmodified records
    const uint32_t *lowPixels; // [lowCount[0] + lowCount[1]] pixel indices, left side's first
    uint32_t lowCount[2];
    unsigned long long *phaseClock; uint32_t clkUnit;   // -DSGZ_DEBUG builds: shader clocks of workgroup `clkUnit` at the phase boundaries
    uint32_t roundSize;       // workgroups that run concurrently, for the XCD-aware order
    uint32_t pipelined;       // 1: other launches run beside this one (sgz_render_queue): the second generation's delayed start and the wave priorities --
                              // tuned for a launch that has the chip to itself -- are left out (tools/pipeline_depth.py: 25.5 -> 22.0 us per render at depth 3)
};
constexpr int kLowBins = 24;
hipError_t launchStftReal(const RealParams &prm, uint32_t N, hipStream_t stream);
// the late pixels of a launch that ran with lateInNext = 1 (pairs), as a launch of its own
hipError_t launchRealLate(const RealParams &prm, uint32_t N, hipStream_t stream);
constexpr int kDecayChunk = 8;    // frames per time chunk of K_B
hipError_t launchStftMap(const StftParams &prm, uint32_t N, int grid, hipStream_t stream);
// the fused kernel's load + three passes only: raw transform Z of every task -> prm.zOut (N = 4096, 32768; Phase mode)
hipError_t launchStftComplex(const StftParams &prm, uint32_t N, int grid, hipStream_t stream);
// N = 2 R^3 (8192, 65536): two workgroups per (frame, pair) write the csf magnitudes of tasks [taskBase, taskBase + grid / 2)
// to prm.binsOut ([task][N + 1]); launchGenericMap turns them into pixels
hipError_t launchStftHalves(const StftParams &prm, uint32_t N, int grid, hipStream_t stream);
// LDS-staged map of the halves path's csf (needs Plan::sideMapOk, binsSplit layout, and mapSidesFit: the arg-max pieces of a
// side fit in LDS beside its N/2 + 48 floats)
bool mapSidesFit(const StftParams &prm, uint32_t N);
hipError_t launchMapSides(const StftParams &prm, uint32_t N, const float *bins, long ntasks, float *mapped, hipStream_t stream);
// after the map kernel: redo prm.dcPixels of every task with the complex csf entries dc[task][kSpecBins] (prm.binsSplit: layout of bins)
hipError_t launchComplexDcFix(const StftParams &prm, uint32_t N, const float *bins, const float2 *dc, long ntasks, float *mapped,
                              hipStream_t stream);
hipError_t launchGenericMap(const StftParams &prm, uint32_t N, const float *bins, long ntasks, float *mapped, hipStream_t stream);
// SpectrumChannels::Phase tables (plan.cpp buildPhaseRecords); null for the other modes
struct PhaseTables {
    const uint32_t *type;     // [P] 0 interpolated, 1 arg-max run, 2 interpolated magnitude only
    const uint32_t *norm;     // [P] normalizedPosition at the pixel's magnitude pass
    uint32_t normFinal;       // normalizedPosition for the arg-max pixels
    uint32_t filtered;        // Linear / Lanczos (cancellation = sqrt(|sum|^2)) vs None (cancellation = |sum| via hypot)
    uint32_t fusedFft;        // N = R^3: the transform comes from launchStftComplex instead of the HBM-resident passes
    float2 *csfOut;           // test hook: complex bins [tasks][N+1] out (or null)
    const float2 *csfIn;      // test hook: map from these complex bins (skip the FFT)
};
hipError_t launchGeneric(const StftParams &prm, uint32_t N, const float2 *twN, float2 *work0, float2 *work1, float *binsWork,
                         long slab, hipStream_t stream, const PhaseTables *phase = nullptr, bool sideMap = false);

struct DecayParams {
    const float *mapped;      // [frames][C][sides][P]
    long frames;
    uint32_t P, C, sides;
    uint32_t chunk;           // frames per time chunk
    uint32_t numChunks;
    const float *slope;       // [P]
    const float *colourTables;// [C][6][3]
    DeviceScalars sc;
    float *agg;               // [numChunks][C][sides][G][P] chunk-end aggregates (zero carry-in, chunk 0 uses state)
    const float *stateIn;     // [C][G][P][2] carry-in (a private copy when numChunks > 1), may be null
    float *stateStash;        // scan launches: copy of every carry-in entry they read (the emit launch reads it from there while the last frame's threads write state), or null
    float *state;             // [C][G][P][2] (float2: left/right) out: state after the last frame, may be null
    uint8_t *rgba;            // [frames][P][4] or null
    float *lines;             // [frames][C][G][P][2] or null
    float magScale;           // factor on the mapped magnitudes before the decay: 1, or 0.5 in Phase mode (mag *= consts::half, TransformDSP.inl:1407) -- fused kernel only
    uint32_t colourOnly;      // neither lines nor state are wanted: only (side 0, LineMain) of every pair feeds the colour column, the scans skip the rest
    uint32_t fusedPixels;     // fused colour kernel: pixels per workgroup (4 -- the fastest launch on an idle device --, 8 or 16: fewer, longer workgroups, less of the chip taken from kernels that run beside it)
    LateFix late; uint32_t hasLate;   // fused colour kernel: `mapped` comes from channel workgroups whose late pixels (late_fix.hpp) are applied while it is read
};
hipError_t launchDecayLocalCarry(const DecayParams &prm, hipStream_t stream);   // local + carry, one launch when the chunks fit a workgroup
hipError_t launchDecayLocal(const DecayParams &prm, hipStream_t stream);
hipError_t launchDecayCarry(const DecayParams &prm, hipStream_t stream);
hipError_t launchDecayEmit(const DecayParams &prm, hipStream_t stream);
// colour column only, one pair, <= 64 time chunks: scan + fold + emit as one launch
bool decayColourFusedApplies(const DecayParams &prm);
hipError_t launchDecayColourFused(const DecayParams &prm, hipStream_t stream);
bool decayFullFusedApplies(const DecayParams &prm);
hipError_t launchDecayFullFused(const DecayParams &prm, hipStream_t stream);
// agg[d] <- max(agg[d], decay(carry)) for aggregates that were scanned from a zero carry-in (multi-GPU carry-apply pass)
hipError_t launchDecayApplyCarry(const DecayParams &prm, const float *carry, hipStream_t stream);
// SpectrumChannels::Phase: sequential-in-time K_B (the cancellation smoother is a linear recurrence: no exact chunk fold);
// work: [frames][C][P] floats for the main graph's dB values
hipError_t launchDecayPhase(const DecayParams &prm, float *work, hipStream_t stream);
hipError_t launchLogf(const float *x, float *y, size_t n, hipStream_t stream);
hipError_t launchFinishPixel(const float *x, float *y, size_t n, hipStream_t stream);   // y = sqrt(x * x + 0) as K_A's pixels compute it   // y = std::log(x) as K_B's dB map computes it (x > 0)
hipError_t launchDecayFold(const float *aggs, const long long *framesPerRank, uint32_t world, uint32_t rank, size_t perRank,
                           uint32_t P, const DeviceScalars &sc, float *carry, hipStream_t stream);

// RSNT (resonator.hip): the resonator bank over `frames` frames of `hop` samples each, starting at planar; state carried in `state`
constexpr int kResSegments = 32;      // segments of frames the chain of a matrix-form launch is cut into (at least 8 frames each)
struct ResParams {
    const float *planar; size_t chStride; long frames;
    uint32_t hop, C, P, mode;
    int V, signals, sides;
    bool firstContinues;              // frame 0 of this launch continues from `state` (always, except inside a long render's later slabs)
    const float2 *coeff;              // [V][P]
    const float4 *cpow;               // [V][P]: pole^hop as (re, im, re_lo, im_lo)
    const float2 *cpowB;              // [V][P][8]: pole^1 .. pole^8
    const float2 *cpowBLo;            // [V][P][2]: low words of pole^4, pole^8
    const float2 *w1, *w2;            // [32][V][P]: the matrix kernels' weights (null: not available for this hop, or switched off)
    const uint4 *w1b;                 // [2 kh][2 h][3 parts][re, im][V][P]: w1 as bf16 parts, eight consecutive samples' weights per entry (B operands)
    int matrixForm;                   // 1: bfloat16 matrix cores, every fp32 value as three exact bf16 parts (opt-in); 2: fp32 matrix cores (default)
    const float4 *tilePow;            // [V][P]: pole^1024 (hi, lo)
    const float *gain;                // [P]
    float weights[9];                 // [V]
    float2 *state;                    // [C][2][V][P]
    float2 *local;                    // [frames][C][signals][V][P]
    float *mapped;                    // [frames][C][sides][P]
    float2 *segEnd;                   // [kResSegments][C][signals][V][P]: the segments' end states from rest (resonatorSegmentKernel), or null
    long segLen;                      // set by the launcher: frames per segment
    bool allFromRest;                 // set by the launcher: every frame of `local` started from rest (matrix kernels), the chain starts at `state`
    bool skipWindow;                  // stop behind the chain (sharded render: the carry of the ranks in front is added first, launchResonatorCarry)
};
hipError_t launchResonator(const ResParams &prm, hipStream_t stream);
// Time-chunk sharding of an RSNT render (sharded.hip).  The recurrence is linear: a rank that starts from rest is short of
// pole^(samples since its chunk began) x (the state that entered its chunk), for every frame.  fold: that entering state from the
// gathered end states of the ranks in front, s <- pole^(chunk_q) s + end_q in fp64 (pole^chunk by squaring pole^hop, hi + lo words);
// carry: local_f += pole^((f + 1) hop) carry for every frame (fp64 walk, one rounding per frame), the plan's state likewise, then the
// window kernel on the corrected states.  carry == null: the window kernel alone (rank 0).
hipError_t launchResonatorFold(const ResParams &prm, const float2 *allEnd /*[world][C][2][V][P]*/, const long long *framesPerRank, uint32_t world,
                               uint32_t rank, float2 *carry /*[C][2][V][P]*/, hipStream_t stream);
hipError_t launchResonatorCarry(const ResParams &prm, const float2 *carry, hipStream_t stream);

}  // namespace sgz
