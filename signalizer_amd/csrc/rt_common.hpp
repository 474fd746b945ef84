// rt_common.hpp -- what the three real-time handles (sgz_spectrum / sgz_scope / sgz_vector) share: a ring of pinned staging slots
// for the audio thread's blocks.  push() never waits for the GPU (SURVEY.md 8(b) "must never block"): a slot whose previous
// upload has not completed is detected with hipEventQuery and the block is refused (SGZ_BUSY) instead of waited for.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

#include "runtime.hpp"

namespace sgz {

struct StageRing {
    static constexpr int kSlots = 8;
    float *h = nullptr;            // pinned  [kSlots][channels][maxBlock]
    float *d = nullptr;            // device  [kSlots][channels][maxBlock]
    hipEvent_t ev[kSlots] = {};
    bool used[kSlots] = {};
    uint32_t channels = 0, maxBlock = 0;
    uint64_t seq = 0;

    sgz_status init(uint32_t nch, uint32_t block)
    {
        release();
        channels = nch; maxBlock = block;
        const size_t bytes = size_t(kSlots) * nch * block * sizeof(float);
        SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&h), bytes, hipHostMallocDefault));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&d), bytes));
        for (int i = 0; i < kSlots; ++i) {
            SGZ_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            used[i] = false;
        }
        seq = 0;
        return SGZ_OK;
    }
    void release()
    {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr;
        for (auto &e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
    // Copies the block into the next pinned slot and enqueues its upload; returns the device copy ([channels][n], row stride n)
    // or nullptr with *st = SGZ_BUSY when the GPU is kSlots blocks behind (nothing waited for, nothing enqueued).
    const float *stage(const float *const *planar, uint32_t n, hipStream_t stream, sgz_status *st)
    {
        const int slot = int(seq % kSlots);
        if (used[slot]) {
            const hipError_t q = hipEventQuery(ev[slot]);
            if (q == hipErrorNotReady) { *st = SGZ_BUSY; return nullptr; }
            if (q != hipSuccess) { *st = hipFail(q, "hipEventQuery"); return nullptr; }
        }
        float *hs = h + size_t(slot) * channels * maxBlock;
        float *ds = d + size_t(slot) * channels * maxBlock;
        for (uint32_t c = 0; c < channels; ++c) std::memcpy(hs + size_t(c) * n, planar[c], size_t(n) * sizeof(float));
        const hipError_t e = hipMemcpyAsync(ds, hs, size_t(channels) * n * sizeof(float), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { *st = hipFail(e, "hipMemcpyAsync"); return nullptr; }
        *st = SGZ_OK;
        return ds;
    }
    // after the kernels that read the slot have been enqueued
    sgz_status commit(hipStream_t stream)
    {
        const int slot = int(seq % kSlots);
        SGZ_HIP(hipEventRecord(ev[slot], stream));
        used[slot] = true;
        ++seq;
        return SGZ_OK;
    }
};

}  // namespace sgz
