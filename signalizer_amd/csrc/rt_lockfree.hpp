// rt_lockfree.hpp -- the host-side lock-free structures of the three real-time handles and the hand-over protocol between the audio
// thread (producer) and the render thread (consumer), WITHOUT a single HIP type: everything in here compiles with a plain C++17
// compiler, so that tests/tsan/rt_lockfree_tsan.cpp can run the very same code under ThreadSanitizer with a mock GPU (round-5 review
// item 5; the structures had one ordering bug found by review and one flaky test, and stress tests on a GPU box cannot see a race).
//
// Threading contract being modelled (reference: the audio thread holds `streamState` for a whole transform, SpectrumDSP.cpp:67, and the
// GL thread takes the same lock, SpectrumRendering.cpp:594; SURVEY.md 8(b) "Threading"): ONE producer thread calls push (never waits),
// ONE consumer thread calls the readers / flush / configure.  What they share:
//   * Backlog        -- host FIFO of blocks the GPU was not ready for: single writer (the producer), single reader (whoever holds the
//                       right to hand blocks to the GPU);
//   * SpinFlag       -- that right for the batched handles (Oscilloscope / Vectorscope): the producer only ever TRIES it;
//   * BatchCore      -- the open batch of a pinned staging slot (touched under the flag only);
//   * the protocol   -- batchPush (producer), batchSync / batchFlushAll (consumer: flush on read), parameterised over the handle's GPU
//                       side (submit / slotReady / gpuIdle / waitGpu): rt_common.hpp's BatchRing supplies the HIP one, the TSAN harness a mock;
//   * ColumnQueue    -- SPSC index pair of the spectrogram's column slots;
//   * LineSeqlock    -- the line results' triple buffer: the producer announces a rewrite before it starts, the consumer re-checks after
//                       its copy.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../include/sgz.h"

namespace sgz {

// Blocks the GPU was not ready for, in arrival order.  One writer, one reader at a time: push() belongs to the producer thread (under the
// handle's push lock); front() / pop() to whoever holds the right to hand blocks to the GPU -- the producer inside its push, or the
// consumer thread in a flush-on-read (scope / vector handles: the batch flag; spectrum handle: the push lock itself) -- so a block parked
// here while the consumer held that right reaches the GPU with the consumer's next read even if no further push ever comes (a stopped
// transport).  Storage is allocated when the handle is configured, never on the audio thread.  Capacity: `seconds` of audio at the handle's
// rate (and at least 32 blocks of the largest push) -- with the 8 staging slots that is how far the GPU may fall behind before audio is lost;
// the reference's stream FIFO has the same kind of bound (its `bufferSize`).
struct Backlog {
    static constexpr int kEntries = 256;
    struct Entry { uint32_t n, channels; size_t off; uint64_t end; };   // end: the write position behind this block (what pop() frees up to)
    float *buf = nullptr;
    size_t cap = 0;
    Entry ent[kEntries];
    // writer
    uint64_t wpos = 0;                                  // floats ever written, the padding skipped at the wraps included
    uint32_t ewr = 0;
    uint64_t deferred = 0;                              // blocks that ever waited here
    // reader
    uint32_t erd = 0;
    std::atomic<uint64_t> rpos{0};                      // everything below this position has been consumed
    std::atomic<uint32_t> count{0};                     // entries waiting (written by both sides: the entry and its samples are published by the increment)

    bool init(size_t floats)                            // false: out of memory
    {
        release();
        buf = static_cast<float *>(std::malloc(floats * sizeof(float)));
        if (!buf) return false;
        cap = floats;
        return true;
    }
    void release() { std::free(buf); buf = nullptr; cap = 0; clear(); }
    void clear() { wpos = 0; ewr = erd = 0; rpos.store(0); count.store(0); }      // (no other thread in the handle: configure / destroy)
    bool push(const float *const *planar, uint32_t channels, uint32_t n)
    {
        const size_t need = size_t(channels) * n;
        if (count.load(std::memory_order_acquire) == uint32_t(kEntries) || need > cap) return false;
        size_t at = size_t(wpos % cap), pad = 0;
        if (at + need > cap) { pad = cap - at; at = 0; }                        // does not fit behind the tail: start over at the front
        if (wpos - rpos.load(std::memory_order_acquire) + pad + need > cap) return false;
        for (uint32_t c = 0; c < channels; ++c) std::memcpy(buf + at + size_t(c) * n, planar[c], size_t(n) * sizeof(float));
        wpos += pad + need;
        ent[ewr % kEntries] = Entry{n, channels, at, wpos};
        ++ewr; ++deferred;
        count.fetch_add(1, std::memory_order_release);
        return true;
    }
    const Entry &front() const { return ent[erd % kEntries]; }                  // (count != 0, read with acquire by the caller's test)
    void pop()
    {
        rpos.store(ent[erd % kEntries].end, std::memory_order_release);
        ++erd;
        count.fetch_sub(1, std::memory_order_release);
    }
};

// The FIFO's size in floats: one second of audio, at least four of the longest blocks -- and never more than 64 MiB however long a block
// the host announces (a 64-channel handle with max_block = 131072 asked for 1 GiB of host memory under the old "32 blocks" rule)
inline size_t backlogFloats(uint32_t channels, double sampleRate, uint32_t maxBlock)
{
    const size_t perChannel = size_t(sampleRate) > size_t(4) * maxBlock ? size_t(sampleRate) : size_t(4) * maxBlock;
    const size_t cap = (size_t(64) << 20) / sizeof(float);
    const size_t want = size_t(channels) * perChannel < cap ? size_t(channels) * perChannel : cap;
    return want > size_t(channels) * maxBlock ? want : size_t(channels) * maxBlock;
}

// push with the FIFO in front: drain what waited (in order) while the GPU takes it, then the new block -- directly if nothing is waiting
// and a slot is free, behind the others otherwise.  pushNow(planar, channels, n) is the handle's own enqueue (SGZ_BUSY = no slot free,
// nothing consumed).  The caller holds the reader's right (see Backlog).
template <typename PushNow>
sgz_status pushThroughBacklog(Backlog &bl, const float *const *planar, uint32_t channels, uint32_t n, PushNow pushNow)
{
    const float *ptrs[64];
    while (bl.count.load(std::memory_order_acquire)) {
        const Backlog::Entry e = bl.front();
        for (uint32_t c = 0; c < e.channels && c < 64; ++c) ptrs[c] = bl.buf + e.off + size_t(c) * e.n;
        const sgz_status st = pushNow(ptrs, e.channels, e.n);
        if (st == SGZ_BUSY) break;
        bl.pop();
        if (st != SGZ_OK) return st;
    }
    if (!bl.count.load(std::memory_order_acquire)) {
        const sgz_status st = pushNow(planar, channels, n);
        if (st != SGZ_BUSY) return st;
    }
    return bl.push(planar, channels, n) ? SGZ_OK : SGZ_BUSY;
}

// Two threads touch the open batch: the producer (append, submit when full) and the consumer (submit on read).  The flag is held around
// every such step; the producer only ever TRIES it (a block that finds it held waits its turn in the Backlog like one that finds no slot
// free), the consumer may spin for the few microseconds an append or an enqueue takes.
struct SpinFlag {
    std::atomic_flag busy = ATOMIC_FLAG_INIT;
    bool tryLock() { return !busy.test_and_set(std::memory_order_acquire); }
    void lock() { while (busy.test_and_set(std::memory_order_acquire)) { } }
    void unlock() { busy.clear(std::memory_order_release); }
};

// The open batch of a staging slot (Oscilloscope, Vectorscope: the ingest kernel takes SEVERAL host blocks per launch): push only copies
// the block behind the ones already waiting in the current pinned slot.  Everything but the flag is touched under the flag only.
// h = [kSlots][channels * slotSamples] floats of host memory (pinned in the product); block b of the open batch = [channels][len[b]] at
// float offset off[b] of slot seq % kSlots.
struct BatchCore : SpinFlag {
    static constexpr int kSlots = 8;
    static constexpr uint32_t kMaxBlocks = 16;
    float *h = nullptr;
    uint32_t channels = 0, slotSamples = 0;
    uint64_t seq = 0;                                   // batches ever committed
    uint32_t count = 0, samples = 0;
    uint32_t off[kMaxBlocks] = {}, len[kMaxBlocks] = {};

    bool fits(uint32_t n) const { return count < kMaxBlocks && samples + n <= slotSamples; }
    float *slotBase() const { return h + size_t(seq % kSlots) * channels * slotSamples; }
    void append(const float *const *planar, uint32_t n)
    {
        float *dst = slotBase() + size_t(channels) * samples;
        for (uint32_t c = 0; c < channels; ++c) std::memcpy(dst + size_t(c) * n, planar[c], size_t(n) * sizeof(float));
        off[count] = channels * samples; len[count] = n;
        samples += n; ++count;
    }
    void committed() { ++seq; count = samples = 0; }    // after the GPU side has enqueued the batch
};

// ---- the hand-over protocol of the batched handles ------------------------------------------------------------------------------
// H (the handle's adapter) provides:
//   BatchCore &batch();  Backlog &backlog();
//   sgz_status submit();       the open batch (count > 0) -> GPU: upload / launch / event, then batch().committed()
//   sgz_status slotReady();    SGZ_OK: the slot of a NEW batch is free; SGZ_BUSY: its previous batch is still in flight (nothing waited for)
//   bool gpuIdle();            has the GPU finished with everything submitted so far?
//   void waitGpu();            wait until it has (never called on the audio thread)
//   bool deferSubmit();        SGZ_RT_OPT_DEFER_SUBMIT

// one block behind the ones already staged (caller holds the batch flag); SGZ_BUSY (nothing consumed) when a new batch would need a
// slot whose last upload is still in flight
template <class H>
sgz_status batchPushNow(H &h, const float *const *blk, uint32_t n)
{
    BatchCore &b = h.batch();
    if (b.count && !b.fits(n))
        if (sgz_status st = h.submit(); st != SGZ_OK) return st;
    if (b.count == 0)
        if (sgz_status st = h.slotReady(); st != SGZ_OK) return st;
    b.append(blk, n);
    // nothing in flight: start now (a busy GPU picks the block up with the next ones).  SGZ_RT_OPT_DEFER_SUBMIT: every block waits for a
    // full batch or a reader -- multi-block launches on demand (tests)
    if (!h.deferSubmit() && h.gpuIdle()) return h.submit();
    return SGZ_OK;
}

// The blocks a push had to park in the host FIFO (the consumer held the batch flag, or no staging slot was free) go behind the open
// batch's, in order.  Caller holds the batch flag and is NOT the audio thread (a staging slot that is still in flight is waited for).
// all = false: only the blocks that wait at the time of the call -- a producer that keeps pushing cannot keep a reader here.
template <class H>
sgz_status batchTakeBacklog(H &h, bool all)
{
    const float *ptrs[64];
    Backlog &bl = h.backlog();
    uint32_t left = bl.count.load(std::memory_order_acquire);
    while (all ? bl.count.load(std::memory_order_acquire) != 0 : left != 0) {
        const Backlog::Entry e = bl.front();
        for (uint32_t c = 0; c < e.channels && c < 64; ++c) ptrs[c] = bl.buf + e.off + size_t(c) * e.n;
        const sgz_status st = batchPushNow(h, ptrs, e.n);
        if (st == SGZ_BUSY) { h.waitGpu(); continue; }
        bl.pop();
        if (left) --left;
        if (st != SGZ_OK) return st;
    }
    return SGZ_OK;
}

// consumer side (flush on read): what waits in the host FIFO and in the open batch goes to the GPU in front of the caller's own work
template <class H>
sgz_status batchSync(H &h)
{
    h.batch().lock();
    sgz_status st = batchTakeBacklog(h, false);
    if (st == SGZ_OK && h.batch().count) st = h.submit();
    h.batch().unlock();
    return st;
}

// sgz_*_flush: everything, also what arrives while this runs (the call may wait: it is not the audio thread's)
template <class H>
sgz_status batchFlushAll(H &h)
{
    h.batch().lock();
    sgz_status st = batchTakeBacklog(h, true);
    if (st == SGZ_OK && h.batch().count) st = h.submit();
    h.batch().unlock();
    return st;
}

// producer side.  Never waits: if the render thread is submitting the open batch right now the block waits its turn in the host FIFO,
// like one the GPU is not ready for; SGZ_BUSY = that FIFO is full.  park = SGZ_RT_OPT_PARK_PUSHES: every block takes that way (the
// tests' handle on a race that timing alone produces).
template <class H>
sgz_status batchPush(H &h, const float *const *planar, uint32_t channels, uint32_t n, bool park)
{
    if (park || !h.batch().tryLock()) return h.backlog().push(planar, channels, n) ? SGZ_OK : SGZ_BUSY;
    auto pushNow = [&](const float *const *blk, uint32_t, uint32_t m) -> sgz_status { return batchPushNow(h, blk, m); };
    const sgz_status st = pushThroughBacklog(h.backlog(), planar, channels, n, pushNow);
    h.batch().unlock();
    return st;
}

// ---- spectrogram columns: SPSC slot queue ----------------------------------------------------------------------------------------
// The producer fills slot tail % Depth (enqueues the copies into it) and bumps tail; the consumer reads slot head % Depth once the
// slot's copy has landed and bumps head.  A full queue drops the column (the reference's frameQueue does: SpectrumDSP.cpp:185-186).
template <int Depth>
struct ColumnQueue {
    std::atomic<uint64_t> head{0}, tail{0};
    void reset() { head.store(0); tail.store(0); }
    // producer
    bool producerSlot(int *slot) const
    {
        const uint64_t t = tail.load(std::memory_order_relaxed);
        if (t - head.load(std::memory_order_acquire) >= uint64_t(Depth)) return false;
        *slot = int(t % Depth);
        return true;
    }
    void producerPublish() { tail.store(tail.load(std::memory_order_relaxed) + 1, std::memory_order_release); }
    // consumer
    uint64_t consumerHead() const { return head.load(std::memory_order_relaxed); }
    bool consumerHas(uint64_t h) const { return h != tail.load(std::memory_order_acquire); }
    void consumerRelease(uint64_t newHead) { head.store(newHead, std::memory_order_release); }
};

// ---- line results: triple buffer with a seqlock --------------------------------------------------------------------------------------
// Copy number n (1, 2, ...) goes into slot (n - 1) % Slots.  The producer ANNOUNCES copy n (`begun`) before it enqueues the transfer and
// PUBLISHES it afterwards; the transfer itself completes later (the GPU's copy engine: `done(slot)` asks its event).  The consumer takes
// the newest published copy whose transfer has completed, copies it out, and checks afterwards that copy n + Slots -- the one that
// rewrites the same slot -- had not been announced: otherwise its read may have overlapped the rewrite and is repeated on a newer copy.
template <int Slots>
struct LineSeqlock {
    std::atomic<uint64_t> begun{0}, published{0};
    void reset() { begun.store(0); published.store(0); }
    static int slotOf(uint64_t n) { return int((n - 1) % Slots); }
    // producer
    uint64_t begin()
    {
        const uint64_t n = begun.load(std::memory_order_relaxed) + 1;
        begun.store(n, std::memory_order_seq_cst);
        return n;
    }
    void publish(uint64_t n) { published.store(n, std::memory_order_release); }
    // consumer.  Returns the copy to read (> 0); 0 with *none = true: nothing has ever been published (the results start zeroed);
    // 0 with *none = false: every candidate is still in flight -- look again.
    template <class Done>
    uint64_t newest(Done done, bool *none) const
    {
        const uint64_t pub = published.load(std::memory_order_acquire);
        uint64_t n = pub;
        while (n > 0 && pub - n < uint64_t(Slots - 1) && !done(slotOf(n))) --n;
        if (n == 0 || !done(slotOf(n))) { *none = pub == 0 || n == 0; return 0; }
        *none = false;
        return n;
    }
    bool stillValid(uint64_t n) const { return begun.load(std::memory_order_seq_cst) < n + uint64_t(Slots); }
};

}  // namespace sgz
