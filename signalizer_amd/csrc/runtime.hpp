// runtime.hpp -- internal helpers shared by api.hip and realtime.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "kernels.hpp"
#include "plan.hpp"

namespace sgz {
extern thread_local std::string g_lastError;
sgz_status fail(sgz_status st, const std::string &msg);
sgz_status hipFail(hipError_t e, const char *what);
#define SGZ_HIP(call)                                        \
    do {                                                     \
        hipError_t _e = (call);                              \
        if (_e != hipSuccess) return ::sgz::hipFail(_e, #call); \
    } while (0)

constexpr size_t kExportGranule = size_t(2) << 20;      // allocations that are exported as dma-buf fds: whole 2 MiB blocks (sgz_export_alloc)
sgz_status ensureCap(float **buf, size_t *cap, size_t need);
int numCUs();
// K_A over `frames` frames (ideal STFT framing from d_planar); any of mapped/binsOut may be null
// deferLate: the caller's next call is runDecayColour on the same d_mapped with only an image wanted -- a channel-split launch may then
// leave its late pixels (late_fix.hpp) to K_B's fused kernel instead of a launch of its own (runDecayColour checks Plan::lateDeferred)
// the plan's second stream and the fork / join events that tie it to the caller's (created on first use, destroyed with the plan)
sgz_status ensureSecondStream(Plan &p);
sgz_status resetResonator(Plan &p, hipStream_t stream);
// RSNT, line-graph mode: the resonators advance over one host block of `nsamples` (sample by sample from the carried state); d_mapped
// [C][sides][P] receives the windowed state afterwards
sgz_status runResonatorAdvance(Plan &p, const float *d_planar, size_t chStride, uint32_t nsamples, float *d_mapped, hipStream_t stream);
// sharded RSNT render (api.hip): this rank's chunk from rest without the window kernel; then the carry of the ranks in front + the windows
sgz_status checkResonatorShardBound(const Plan &p, long frames);
sgz_status runResonatorFromRest(Plan &p, const float *d_planar, size_t chStride, long frames, float *d_mapped, hipStream_t stream);
sgz_status runResonatorJoin(Plan &p, long frames, float *d_mapped, const float *d_allEnd, const long long *framesPerRank, uint32_t world, uint32_t rank,
                            float *d_carry, hipStream_t stream);
sgz_status runStft(Plan &p, const float *d_planar, size_t chStride, long frames, float *d_mapped,
                   float *d_binsOut, const float *d_binsIn, hipStream_t stream, unsigned long long *d_phaseClock = nullptr,
                   bool deferLate = false);
#ifdef SGZ_DEBUG
extern uint32_t g_ablate;   // debug only (tools/ablate.py)
#endif
// K_B: decay recurrence + dB map + colour blend
sgz_status runDecayColour(Plan &p, const float *d_mapped, long frames, uint8_t *d_rgba, float *d_lines,
                          float *d_state, hipStream_t stream, bool magnitudeOnly = false);
sgz_status runDecayEmitWithCarry(Plan &p, const float *d_mapped, long frames, const float *d_carry, uint8_t *d_rgba, float *d_lines,
                                 float *d_stateOut, hipStream_t stream);
// frequency tracker (tracker.hip): peak search + parabolic fit on one (frame, pair)'s csf magnitudes; d_out: DEVICE sgz_peak
sgz_status runTrackPeak(const Plan &p, const float *d_bins, double mouseFraction, sgz_peak *d_out, hipStream_t stream);
sgz_status trackPeakLines(const Plan &p, const float *results /*host float2 [P]*/, double mouseFraction, sgz_line_peak *out);   // tracker.hip
}  // namespace sgz
