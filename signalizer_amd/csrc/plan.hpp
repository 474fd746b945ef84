// plan.hpp -- host-side "constant block" of the Spectrum path (the MI355X counterpart of
// Signalizer::TransformConstant<float>, Source/Spectrum/TransformConstant.h:48-241) and the device
// tables derived from it.  Built on the CPU in fp64 with the reference's expression order, uploaded once.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sgz.h"

namespace sgz {

constexpr int kMaxTaps = 10;   // Lanczos a=5 => 10 taps (TransformDSP.inl:514, lanczosFilterSize = 5)

// One record per (side, pixel): how mapToLinearSpace produces csp[x] (TransformDSP.inl:565-639, :878-984).
struct PixelRec {
    int32_t kind;    // 0 = interpolate (None/Linear/Lanczos taps); odd = arg-max of |X|^2 over a bin run, kind >> 1 = index
                     // of the run's first <=16-bin piece in the MaxItem list
    int32_t a;       // interp: first tap index into csf (already wrapped into [0,N]); max: first offset
    int32_t b;       // interp: number of taps;                                           max: run length (>=1)
    int32_t c;       // interp: offset of this pixel's weights in the weight table;         max: fallback bin
};

// Work item ("piece") of the balanced arg-max: the part of one record's bin run that falls into one 16-aligned
// window of csf indices k in [16 w, 16 w + 16); positions lo..hi (inclusive) of the window belong to the run.
// Pieces of a run are listed in the reference's scan order (ascending offset: ascending k on the left side,
// descending k on the right side, where k = N - offset).
struct MaxItem {
    uint32_t win;       // w | lo << 16 | hi << 20   (one word per piece: every workgroup of a launch reads the whole list)
};

// Chunk-scan pixel map (chunk_map.hpp; tables from plan.cpp buildChunkMap).  A side's magnitudes S[i], i = 0 .. M (left: csf[i],
// right: csf[M + i]) live in LDS at float position chunkPos(i): two pad floats after every 32 entries, so that a thread's chunk of 32
// is 8-byte aligned (ds_read_b64) and both the strided stores of the transform and the chunk reads are free of bank conflicts.
constexpr int chunkPos(int i) { return i + ((i >> 5) << 1); }
constexpr int kTapFloats = kMaxTaps + 2;          // a tap window as contiguous floats: <= 10 taps + the 2 pad slots it may step over
// ChunkRec = two words per (side, pixel):
//   interpolated pixel:  [0] = row of its weights in weights12,  [1] = float position of its first tap | kChunkInterp
//   arg-max pixel:       [1] = chunks before the last one that its run covers | flags << 16;
//                        [0] = index of the run's tile maximum | first of those chunks << 16   (kChunkDirect: float position of the run's one entry)
constexpr uint32_t kChunkDirect = 1u << 16;       // the run has one entry inside the chunks: read it
constexpr uint32_t kChunkPlusM = 1u << 17;        // the run includes entry M (csf[N/2] of the left side / a mono signal), which no chunk holds
constexpr uint32_t kChunkNoScan = 1u << 18;       // the run has no entry inside the chunks
constexpr uint32_t kChunkInterp = 1u << 19;       // [1]: an interpolated pixel (its float position in the low 16 bits)
constexpr uint32_t kChunkOff = 1u << 20;          // [1]: nothing to map here (a pixel settled elsewhere: realLowPixels; or no pixel at all)

// Scalars the kernels need (all derived on the host exactly as the reference derives them).
struct DeviceScalars {
    float invSize;        // windowKernelScale / (W/2), TransformDSP.inl:540
    float pole[SGZ_NUM_GRAPHS];
    float phasePole[SGZ_NUM_GRAPHS];   // pole^0.3, the one-pole smoother of Phase mode's cancellation (TransformDSP.inl:1397-1398)
    float deltaYRecip;    // 1 / ln(upper/lower), TransformDSP.inl:1311
    float minFracRecip;   // 1 / lower,           :1312
    float lowerClip;      // (float)clipDB,       :1314
    float ratios[SGZ_NUM_SPEC_COLOURS + 1];   // normalizedSpecRatios, Spectrum.cpp:226-246
};

struct Plan {
    sgz_spectrum_config cfg{};
    uint32_t N = 0, W = 0, P = 0, C = 0;
    int log2N = 0;
    int sides = 1;              // 2 for Separate/MidSide (left at csp[0,P), right at csp[P,2P))
    int stateChannels = 1;      // TransformConstant::getStateConfigurationChannels, TransformConstant.h:183-186
    double windowScale = 1.0;   // windowKernelScale
    uint32_t breakPixel = 0;

    // host tables
    std::vector<float> window;          // N entries, [W,N) zero   (windowKernel)
    std::vector<float> mapped;          // P   (mappedFrequencies)
    std::vector<float> slope;           // P   (slopeMap)
    std::vector<float> colourTables;    // C * 6 * 3 (generateSpectrogramColourRotation per pair)
    std::vector<PixelRec> recs;         // sides * P
    std::vector<float> weights;         // packed tap weights
    std::vector<PixelRec> recsReal;     // channel-split kernels: recs with the pixels of realLowPixels switched off (empty: recs itself)
    std::vector<uint32_t> realLowPixels; // pixels whose taps reach over bin 0 into the other channel's entries: left side's, then right side's
    uint32_t realLowCount[2] = {0, 0};
    std::vector<float> weights11;       // N = R^3 (fused kernel): [record][kMaxTaps + 1], see WholeFrameIndex::kLinearTaps
    std::vector<uint32_t> phaseType, phaseNorm;   // Phase mode only (plan.cpp buildPhaseRecords), [P] each
    uint32_t phaseNormFinal = 0;
    std::vector<MaxItem> items;         // arg-max runs cut into <= 16-bin pieces (left-side records first)
    uint32_t nItemsLeft = 0;
    std::vector<float> tw1, tw2;        // FFT twiddles (re,im interleaved), see fft kernels (fused N = R^3 path)
    std::vector<float> twN;             // generic path: W_N^i, i < N/2
    std::vector<uint32_t> dcPixels;     // Complex mode: pixels whose taps / arg-max run include csf[0] (kept complex, TransformDSP.inl:993)
    std::vector<float> tw1odd;          // halves path: pass-1 twiddles of the odd half
    bool fused = false;                 // N in {4096, 32768}: spectrum_fft.hip; otherwise spectrum_generic.hip
    bool phaseFusedFft = false;         // Phase at N = R^3: the transform comes from the in-register FFT (complex output), the rest is generic
    bool sideMapOk = false;             // halves path: every record stays inside the csf range mapSideKernel stages per side
    bool halves = false;                // N in {8192, 65536} = 2 R^3: two half-frame workgroups (spectrum_fft.hip) + genericMap
    // Channel-split path (spectrum_real.hip): Separate mode at N = 32768 / 65536 with W == N -- every (frame, pair, channel) is its own
    // workgroup: a real-input FFT of N/2 complex points, that channel's csf side and that side's pixels.  Eligible when every record
    // of a side stays inside that side's half of csf (no wrap-around taps) and a side's arg-max pieces fit beside |X| in LDS.
    bool realSplit = false;
    bool realMono = false;              // Left / Right / Merge / Side on the real-input kernel, one workgroup per (frame, pair)
    std::vector<float> twReal1;         // pass-1 twiddles W_{N/2}^{c q}, factorised rows [3 + R1/4 - 1][1024] (re, im)
    uint32_t realFixFrom[2] = {0, 0};   // per side: first pixel whose arg-max run ends on csf[N/2] (the one cross-channel entry); [from, P) are settled late
    std::vector<float> winPhaseT;       // fused whole-frame kernel, same idea: (cos, sin) of 2 pi t / N, t < R^2
    std::vector<float> windowHalf;      // channel-split path: window / 2 (those kernels transform x w / 2: real_common.hpp realBinMag)
    std::vector<float> winPhase; float winP0 = 0.f, winP1 = 0.f;   // channel-split path, Hann / Hamming periodic: the window is computed in the kernel
    std::vector<float> twRealPost;      // W_N^{kc}, kc < R1 * 32: the real-FFT recombination twiddle of a thread's bins
    std::vector<float> tw16, twPost16;  // the 1024-thread form of the N = 32768 channel transform (spectrum_real16.hip): pass-2 / pass-3 twiddles (LDS-staged), W_N^{kb}
    std::vector<float> tw2Full;         // channel-split kernels, N >= 32768: the whole pass-2 table W_1024^{c q} as [c < 32][34] float2 (re, im), q < 32 used, two pad entries per row (a thread's 32 factors are 16 ds_read_b128); staged in LDS
    // Chunk-scan pixel map of the channel-split kernels (chunk_map.hpp, built by buildChunkMap): a side's M magnitudes are cut into
    // T chunks of 32 consecutive entries, one per thread; the arg-max runs of >= 2 entries ("tiles") are segments of a segmented
    // running maximum inside the chunks.
    std::vector<uint32_t> chunkEnds;    // [side][T]: bit j = the thread's chunk element j is the last entry of a tile (or sits right before one)
    std::vector<uint32_t> chunkReBase;  // [side][T]: number of such entries below the thread's chunk = index of its first tile maximum
    std::vector<uint32_t> chunkRec;     // [side][P][2]: per pixel, see ChunkRec in chunk_map.hpp
    std::vector<float> weights12;       // [interpolated pixel][12]: its taps as 12 contiguous floats of the padded array (0 on pad slots and behind the last tap)
    uint32_t chunkSlots[2] = {0, 0};    // tile maxima per side
    // RSNT (resonator.hip; CComplexResonator::Constant restated): V = 2K - 1 detuned resonators per axis point for a K-term cosine-sum window
    int resV = 0;
    std::vector<float> resCoeff;        // [V][P] (re, im): pole r e^{j w}
    std::vector<float> resPow;          // [V][P] (re, im, re_lo, im_lo): (the fp32 pole)^hop, evaluated in double, as hi + lo fp32 words -- the carry of a whole frame
    std::vector<float> resPowB;         // [V][P][8] (re, im): (the fp32 pole)^1 .. ^8, rounded once each -- the block steps of resonateKernel
    std::vector<float> resPowBLo;       // [V][P][2] (re, im): low words of pole^4 and pole^8
    std::vector<float> resW1, resW2;    // hop % 1024 == 0: [32][V][P] (re, im): pole^(31 - b) and pole^(32 (31 - a)), the weights of resonateMfmaKernel (resonator index last: a wave's 32 lanes read 32 neighbours)
    std::vector<uint32_t> resW1b;       // [2][2][3][2][V][P][4]: resW1 as three bfloat16 parts each (x = h + m + l exactly), packed as the B operands of resonateMfmaBf16Kernel
    std::vector<float> resTile;         // [V][P] (re, im, re_lo, im_lo): pole^1024
    std::vector<float> resGain;         // [P]
    float resWeights[9] = {0};          // [V]
    DeviceScalars scalars{};
    // sgz_plan_set_option
    bool optChannelSplit = true, optFusedColour = true, optFetchWindow = false, optWideGroups = false;
    bool optPipelined = false;          // the plan is a lane of an sgz_render_queue of depth >= 2 (RealParams::pipelined)
    uint32_t optFusedPixels = 4;        // pixels per workgroup of the fused colour K_B (4, 8, 16): SGZ_OPT_FUSED_COLOUR = 1 / 8 / 16
    int optMatrixResonator = 2;        // 0: vector ALUs, 1: bf16 matrix cores (three-part split; opt-in: on MI355X its instruction stream disturbs FFT
                                       //    kernels running beside it -- rocFFT's too --, NOTES.md round 6), 2: fp32 matrix cores (default since round 6)
    uint32_t optResonatorSlab = 0;      // RSNT: frames per slab of a long render (0: as many as fit 256 MiB of per-frame states)
    uint32_t optResonatorShardBound = 0; // RSNT, sharded render: most frames a rank's chunk may hold (0: 8 GiB worth of per-frame states)

    // device mirrors (owned)
    bool uploaded = false;
    float *d_window = nullptr, *d_slope = nullptr, *d_colourTables = nullptr, *d_weights = nullptr, *d_weights11 = nullptr;
    float *d_tw1 = nullptr, *d_tw2 = nullptr, *d_twN = nullptr, *d_tw1odd = nullptr;
    float *d_work0 = nullptr, *d_work1 = nullptr, *d_binsWork = nullptr; size_t workSlab = 0;
    uint32_t *d_dcPixels = nullptr; float *d_dcWork = nullptr; size_t dcSlab = 0;   // Complex mode: pixel list, csf[0] of a slab of tasks
    float *d_halfBins = nullptr; size_t binsSlab = 0;                                            // halves path: csf magnitudes of a slab of tasks   // generic path buffers
    uint32_t *d_phaseType = nullptr, *d_phaseNorm = nullptr;
    PixelRec *d_recs = nullptr;
    MaxItem *d_items = nullptr;
    // work buffers (grown on demand)
    float *d_mapped = nullptr; size_t mappedCap = 0;      // [frames][pairs][2][P]
    float *d_agg = nullptr; size_t aggCap = 0;            // decay chunk aggregates
    const float *aggMapped = nullptr; long aggFrames = 0; // whose zero-carry scan d_agg holds (sgz_stage_decay_emit checks it), null: nobody's
    float *d_stateCopy = nullptr; size_t stateCopyCap = 0;  // carry-in snapshot (decayEmit reads it while writing state)
    float *d_phaseWork = nullptr; size_t phaseWorkCap = 0;   // Phase mode: main-graph dB values [frames][C][P]
    PixelRec *d_recsReal = nullptr; uint32_t *d_realLowPixels = nullptr; float *d_low = nullptr;
    float *d_scratch = nullptr; size_t scratchCap = 0;    // per-workgroup bin scratch (N > 32768)
    // sgz_spectrogram_render_host: device copies of the caller's host buffers, the stream they move on, timing events
    float *d_hostAudio = nullptr, *d_hostRgba = nullptr, *d_hostLines = nullptr;
    size_t hostAudioCap = 0, hostRgbaCap = 0, hostLinesCap = 0;
    void *hostStream = nullptr;                           // hipStream_t / hipEvent_t (this header is also compiled as plain C++)
    void *hostEv[4] = {nullptr, nullptr, nullptr, nullptr};
    float *d_tw2Full = nullptr;
    float *d_tw16 = nullptr, *d_twPost16 = nullptr, *d_windowHalf = nullptr;
    float *d_twReal1 = nullptr, *d_twRealPost = nullptr, *d_winPhase = nullptr, *d_winPhaseT = nullptr;
    uint32_t *d_chunkEnds = nullptr, *d_chunkReBase = nullptr, *d_chunkRec = nullptr; float *d_weights12 = nullptr;
    float *d_ny = nullptr, *d_nyBest = nullptr; size_t nyCap = 0;   // channel-split path: what a frame's two channel workgroups leave for realLateKernel
    const float *lateDeferred = nullptr;   // the `mapped` buffer whose channel-split K_A left its late pixels (late_fix.hpp) to the next K_B on it, or null
    long lateFrames = 0;                   // ... and how many frames that launch covered (d_ny / d_nyBest hold exactly those)
    void *shardStream = nullptr; void *shardEv[2] = {nullptr, nullptr};   // sgz_spectrogram_render_sharded: the halo exchange's own stream (hipStream_t / hipEvent_t)
    uint32_t *d_resW1b = nullptr;
    float *d_resCoeff = nullptr, *d_resPow = nullptr, *d_resPowB = nullptr, *d_resPowBLo = nullptr, *d_resW1 = nullptr, *d_resW2 = nullptr, *d_resTile = nullptr, *d_resGain = nullptr;
    float *d_resState = nullptr;                          // [C][2][V][P] (re, im): the resonators between calls
    float *d_resLocal = nullptr; size_t resLocalCap = 0;  // [frames][C][signals][V][P] (re, im): per-frame sums from rest
    float *d_shard = nullptr; size_t shardCap = 0;        // sgz_spectrogram_render_sharded: end state, carry, gathered states, halo packs
    int device = 0;

    ~Plan();
};

inline bool isResonator(const Plan &p) { return p.cfg.algorithm == SGZ_ALGO_RSNT; }
inline long planFrames(const Plan &p, size_t nsamples)
{
    return isResonator(p) ? long(nsamples / p.cfg.hop) : long(sgz_num_frames(nsamples, p.W, p.cfg.hop));
}

// Builds every host table; returns SGZ_OK or an error (message in `err`).
sgz_status buildPlan(const sgz_spectrum_config &cfg, Plan &plan, std::string &err);
sgz_status uploadPlan(Plan &plan, std::string &err);

void rotateHueRgb8(const uint8_t rgb[3], float amount, uint8_t out[3]);
double designWindow(uint32_t type, uint32_t symmetry, double alpha, double beta, uint32_t W, float *out);
uint32_t transformSizeFor(uint32_t W);
double lanczosKernel(double d, int a);

}  // namespace sgz
