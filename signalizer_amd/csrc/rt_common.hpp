// rt_common.hpp -- what the three real-time handles (sgz_spectrum / sgz_scope / sgz_vector) share: a ring of pinned staging slots
// for the audio thread's blocks, and a host FIFO in front of it.  push() never waits for the GPU (SURVEY.md 8(b) "must never
// block"): a slot whose previous upload has not completed is detected with hipEventQuery, and the block then WAITS ITS TURN in the
// FIFO (Backlog below) -- the stream the kernels see has no holes, as the reference's cpl::AudioStream FIFO guarantees short of its own
// overflow (PluginProcessor.cpp:195-198, MixGraphListener.cpp:336-387).  Only a full FIFO refuses a block (SGZ_BUSY).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "rt_lockfree.hpp"     // Backlog, SpinFlag, BatchCore, the hand-over protocol, ColumnQueue, LineSeqlock: HIP-free (ThreadSanitizer harness)
#include "runtime.hpp"

namespace sgz {

// Is `p` memory the DMA engines can write directly (hipHostMalloc / hipHostRegister, i.e. pinned)?  A plain malloc'd pointer makes
// hipPointerGetAttributes fail: the sticky error is cleared.
inline bool isPinnedHost(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// Can kernels on `stream` write memory that lives on device `owner`?  Yes on the stream's own device; on another one only with peer
// access (enabled here if the link allows it -- "already enabled" is fine).  Anything else would fault inside the vertex kernel.
inline bool streamCanWriteDevice(hipStream_t stream, int owner)
{
    hipDevice_t mine = 0;
    if (hipStreamGetDevice(stream, &mine) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (int(mine) == owner) return true;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, int(mine), owner) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
    int keep = 0;
    (void)hipGetDevice(&keep);
    if (hipSetDevice(int(mine)) != hipSuccess) { (void)hipGetLastError(); return false; }
    const hipError_t e = hipDeviceEnablePeerAccess(owner, 0);
    (void)hipGetLastError();
    (void)hipSetDevice(keep);
    return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
}

// The address under which a kernel on `stream` can write `p` directly: `p` itself for DEVICE memory of the stream's device or of a peer
// it has access to (a mapped vertex buffer object, exported memory the display GPU imported: the vertices stay in HBM), or the
// device-side alias of pinned host memory that is mapped into the device's address space (hipHostMalloc, hipHostRegister with
// hipHostRegisterMapped; torch's pin_memory): the vertex kernels then store straight into the caller's buffers over PCIe -- no
// device-side staging, no DMA copy behind the kernel.  nullptr otherwise (pageable memory, another GPU's memory without peer access:
// those go through the staging path, whose copies the runtime routes).
inline void *mappedDevicePointer(const void *p, hipStream_t stream = nullptr)
{
#ifdef SGZ_NO_DIRECT_HOST_WRITES
    (void)p; (void)stream; return nullptr;
#else
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type == hipMemoryTypeDevice)
        return ((reinterpret_cast<uintptr_t>(p) & 3) || !streamCanWriteDevice(stream, at.device)) ? nullptr : const_cast<void *>(p);
    if (at.type != hipMemoryTypeHost || !at.devicePointer || (reinterpret_cast<uintptr_t>(at.devicePointer) & 3)) return nullptr;
    return at.devicePointer;
#endif
}

// true when a copy command can write `p` (pinned host memory or device memory: the DMA engine / the runtime's peer path reaches it)
inline bool isCopyTarget(const void *p)
{
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice;
}

// Device results -> the caller's buffers, then wait.  Each destination on its own: pinned host memory and DEVICE memory (a caller may
// hand a device buffer for one part and a pageable one for the other) are written by the copy itself; pageable destinations go through
// the handle's pinned bounce buffer `h_bounce` (room for both parts) and a host copy.
inline sgz_status readBack(void *dstA, const void *d_a, size_t bytesA, void *dstB, const void *d_b, size_t bytesB, void *h_bounce,
                           hipStream_t stream)
{
    const bool directA = isCopyTarget(dstA), directB = dstB && isCopyTarget(dstB);
    char *hb = static_cast<char *>(h_bounce);
    SGZ_HIP(hipMemcpyAsync(directA ? dstA : hb, d_a, bytesA, hipMemcpyDefault, stream));
    if (dstB) SGZ_HIP(hipMemcpyAsync(directB ? dstB : hb + bytesA, d_b, bytesB, hipMemcpyDefault, stream));
    SGZ_HIP(hipStreamSynchronize(stream));
    if (!directA) std::memcpy(dstA, hb, bytesA);
    if (dstB && !directB) std::memcpy(dstB, hb + bytesA, bytesB);
    return SGZ_OK;
}

struct StageRing {
    static constexpr int kSlots = 8;
    float *h = nullptr;            // pinned  [kSlots][channels][maxBlock]
    float *hd = nullptr;           // the same memory as the device sees it (null: not mapped)
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
        hd = static_cast<float *>(mappedDevicePointer(h));
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
    // Copies the block into the next pinned slot; returns what the ingest kernel reads ([channels][n], row stride n): the slot itself
    // as the device sees it (every sample is read once: no copy command in front of the launch), or its uploaded twin when the slot
    // is not mapped -- or nullptr with *st = SGZ_BUSY when the GPU is kSlots blocks behind (nothing waited for, nothing enqueued).
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
        *st = SGZ_OK;
        if (hd) return hd + size_t(slot) * channels * maxBlock;
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

// Staging for handles whose ingest kernel takes SEVERAL host blocks per launch (Oscilloscope, Vectorscope): push only copies the block
// behind the ones already waiting in the current pinned slot; the upload and the kernel are enqueued at once when the GPU has finished
// the previous batch (an idle GPU starts on a block straight away, as before) and otherwise when somebody needs the result -- the render
// thread's calls (flush on read), a full slot, flush().  A GPU that is behind therefore catches up with ONE staged copy and ONE launch
// for everything that arrived meanwhile, instead of one of each per audio callback.  Blocks keep their boundaries: the kernel runs the
// per-callback state machine over them in order.  (Measured, round 4: deferring EVERY submission to the render thread's read -- one
// launch per rendered frame -- made BASELINE configs[2] / [3] slower, 0.43 -> 0.61 ms and 0.19 -> 0.41 ms per frame: the ingest kernel
// costs ~20 us per block whether the blocks come in one launch or seven, and what the eager form overlaps with the audio thread's next
// callbacks the deferred form runs while the render thread waits.)
//
// The open batch itself, the flag two threads take around it and the hand-over protocol are rt_lockfree.hpp's (BatchCore, batchPush /
// batchSync / batchFlushAll); this struct adds the GPU side of a slot: its device twin, its event, the upload.
struct BatchRing : BatchCore {
    float *hd = nullptr;           // the pinned slots (BatchCore::h) as the device sees them (null: not mapped -- the batch is copied by the DMA engine)
    float *d = nullptr;            // device  [kSlots][channels * slotSamples]
    hipEvent_t ev[kSlots] = {};
    bool used[kSlots] = {};

    sgz_status init(uint32_t nch, uint32_t samplesPerSlot)
    {
        release();
        // (a multiple of four samples: every slot base -- slot x channels x slotSamples floats -- is then 16-byte aligned for batchFetch's
        // float4 accesses whatever max_block and the channel count are)
        samplesPerSlot = (samplesPerSlot + 3u) & ~3u;
        channels = nch; slotSamples = samplesPerSlot;
        const size_t bytes = size_t(kSlots) * nch * samplesPerSlot * sizeof(float);
        SGZ_HIP(hipHostMalloc(reinterpret_cast<void **>(&h), bytes, hipHostMallocDefault));
        SGZ_HIP(hipMalloc(reinterpret_cast<void **>(&d), bytes));
        hd = static_cast<float *>(mappedDevicePointer(h));
        for (int i = 0; i < kSlots; ++i) {
            SGZ_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            used[i] = false;
        }
        seq = 0; count = samples = 0;
        return SGZ_OK;
    }
    void release()
    {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr;
        for (auto &e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        count = samples = 0;
    }
    // has the GPU finished with everything submitted so far?  (then a new block may as well start now: batching is for a GPU that is
    // behind, not a reason to let an idle one wait for the render thread)
    bool idle()
    {
        if (seq == 0) return true;
        const int slot = int((seq - 1) % kSlots);
        return !used[slot] || hipEventQuery(ev[slot]) == hipSuccess;
    }
    // SGZ_OK: the slot of a NEW batch is free; SGZ_BUSY: its previous upload / kernel is still in flight (nothing waited for)
    sgz_status slotReady()
    {
        const int slot = int(seq % kSlots);
        if (!used[slot]) return SGZ_OK;
        const hipError_t q = hipEventQuery(ev[slot]);
        if (q == hipErrorNotReady) return SGZ_BUSY;
        if (q != hipSuccess) return hipFail(q, "hipEventQuery");
        used[slot] = false;
        return SGZ_OK;
    }
    // the open batch -> device; returns the device address of its first block.  When the pinned slot is mapped into the device's address
    // space the ingest kernel fetches the batch itself (batchFetch below: *fetchFrom = the slot as the device sees it, `floats` values):
    // a copy command in front of every ingest launch cost 5 us of engine time and 10-25 us of gaps around it.  Otherwise one DMA copy.
    const float *upload(hipStream_t stream, sgz_status *st, const float **fetchFrom, uint32_t *floats)
    {
        const int slot = int(seq % kSlots);
        const size_t at = size_t(slot) * channels * slotSamples;
        *st = SGZ_OK;
        *floats = channels * samples;
        if (hd) { *fetchFrom = hd + at; return d + at; }
        *fetchFrom = nullptr;
        const hipError_t e = hipMemcpyAsync(d + at, h + at, size_t(channels) * samples * sizeof(float), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { *st = hipFail(e, "hipMemcpyAsync"); return nullptr; }
        return d + at;
    }
    // after the kernel that reads the batch has been enqueued
    sgz_status commit(hipStream_t stream)
    {
        const int slot = int(seq % kSlots);
        SGZ_HIP(hipEventRecord(ev[slot], stream));
        used[slot] = true;
        committed();
        return SGZ_OK;
    }
};

// first thing in an ingest kernel (ONE workgroup): the staged batch from the mapped pinned slot into its HBM twin, every thread 16 bytes
// at a time (one PCIe round trip for the lot); the phases behind it read the HBM copy as before.  (Slot bases are multiples of four
// floats: channels x slotSamples, slotSamples >= 8192.)
#ifdef __HIPCC__
__device__ __forceinline__ void batchFetch(const float *from, float *to, uint32_t floats, int tid, int threads)
{
    if (!from) return;                                       // (uniform: the host copied)
    const uint32_t quads = floats / 4u;
    // Every read is a round trip over PCIe (~2 us): eight of a thread's reads are in flight before the first is stored, so that a batch of
    // up to 8 x threads x 16 bytes costs ONE round trip (as a plain copy loop the compiler keeps one load in flight per thread: a
    // 512-sample callback of 8 channels on 256 threads was four round trips in a row -- round 6)
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *src = reinterpret_cast<const v4f *>(from);
    v4f *dst = reinterpret_cast<v4f *>(to);
    const uint32_t T = uint32_t(threads);
    for (uint32_t q0 = uint32_t(tid); q0 < quads; q0 += 8u * T) {
        // (unconditional loads from clamped indices, conditional stores; eight named values: an array here went to scratch)
        const uint32_t last = quads - 1u;
        const uint32_t i0 = q0, i1 = q0 + T, i2 = q0 + 2u * T, i3 = q0 + 3u * T, i4 = q0 + 4u * T, i5 = q0 + 5u * T, i6 = q0 + 6u * T, i7 = q0 + 7u * T;
        const v4f a0 = src[i0 < last ? i0 : last], a1 = src[i1 < last ? i1 : last], a2 = src[i2 < last ? i2 : last], a3 = src[i3 < last ? i3 : last];
        const v4f a4 = src[i4 < last ? i4 : last], a5 = src[i5 < last ? i5 : last], a6 = src[i6 < last ? i6 : last], a7 = src[i7 < last ? i7 : last];
        if (i0 < quads) dst[i0] = a0;
        if (i1 < quads) dst[i1] = a1;
        if (i2 < quads) dst[i2] = a2;
        if (i3 < quads) dst[i3] = a3;
        if (i4 < quads) dst[i4] = a4;
        if (i5 < quads) dst[i5] = a5;
        if (i6 < quads) dst[i6] = a6;
        if (i7 < quads) dst[i7] = a7;
    }
    for (uint32_t e = quads * 4u + uint32_t(tid); e < floats; e += uint32_t(threads)) to[e] = from[e];
    __threadfence_block();
    __syncthreads();
}
#endif

}  // namespace sgz
