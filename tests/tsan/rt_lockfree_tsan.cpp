// rt_lockfree_tsan.cpp -- ThreadSanitizer harness for signalizer_amd/csrc/rt_lockfree.hpp (round-5 review item 5).
//
// The header under test is the product's own: Backlog, SpinFlag / BatchCore, the hand-over protocol (batchPush / batchSync /
// batchFlushAll / batchTakeBacklog / pushThroughBacklog), ColumnQueue and LineSeqlock are compiled here exactly as libsgz.so compiles
// them; only the GPU is a mock -- an executor thread that runs enqueued commands in order (a stream) and completes mock events (the
// latest record decides, as with hipEventQuery).  Threads: ONE producer (the audio thread: push, never waits, re-offers a refused block),
// ONE consumer (the render thread: flush-on-read, pop_column, line_results, flush), ONE control thread (set_option / configure: by the
// library's contract configure comes from the consumer side, so it shares the host's consumer-side lock with the reads; what the LIBRARY
// must get right is configure against a concurrent push -- the handle mutex the producer only ever tries).
//
// Two kinds of checking: ThreadSanitizer watches every plain-memory hand-over (staging slots, the FIFO's buffer and entries, column slots:
// all plain memory on purpose), and the mock GPU / the consumer check the LOGIC -- blocks reach the GPU complete, in order, exactly once,
// parked blocks reach it with the next read although no further push comes, columns arrive in order and unmixed, a line-results read is
// never torn.  (The triple buffer's payload is relaxed atomics: a seqlock READS data that may be rewritten and discards it afterwards,
// which is a data race by the letter of the C++ model whatever the protocol does; in the product the writer is the copy engine.)
//
// usage: rt_lockfree_tsan [blocks_per_scenario]      exit 0 and a line "tsan harness ok: N operations" on success
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../signalizer_amd/csrc/rt_lockfree.hpp"

using namespace sgz;

static std::atomic<uint64_t> g_ops{0};
// (relaxed: a sequentially consistent counter shared by all threads would itself order them and hide the very races this harness looks for)
static inline void opTick() { g_ops.fetch_add(1, std::memory_order_relaxed); }
#define CHECK(cond)                                                                       \
    do {                                                                                  \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); std::abort(); } \
    } while (0)

// ---- the mock GPU --------------------------------------------------------------------------------------------------------------
struct MockEvent {                                       // hipEventRecord / hipEventQuery: the most recent record decides
    std::atomic<uint64_t> recorded{0}, completed{0};
    uint64_t record() { return recorded.fetch_add(1, std::memory_order_relaxed) + 1; }
    bool done() const { return completed.load(std::memory_order_acquire) == recorded.load(std::memory_order_relaxed); }
    bool everRecorded() const { return recorded.load(std::memory_order_relaxed) != 0; }
};

struct MockStream {                                      // commands run in order on one thread, like a HIP stream
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::atomic<uint64_t> enqueued{0}, executed{0};
    std::atomic<uint32_t> slowness{0};                   // microseconds of sleep per command now and then (a GPU that falls behind)
    bool stop = false;
    std::thread th;
    MockStream() : th([this] { run(); }) {}
    ~MockStream()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        th.join();
    }
    void enqueue(std::function<void()> f)
    {
        { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); }
        enqueued.fetch_add(1, std::memory_order_relaxed);
        cv.notify_one();
    }
    void record(MockEvent &e)
    {
        const uint64_t ticket = e.record();
        enqueue([&e, ticket] { e.completed.store(ticket, std::memory_order_release); });
    }
    void synchronize() { const uint64_t want = enqueued.load(std::memory_order_relaxed); while (executed.load(std::memory_order_acquire) < want) std::this_thread::yield(); }
    void run()
    {
        std::minstd_rand rng(7);
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            const uint32_t s = slowness.load(std::memory_order_relaxed);
            if (s && rng() % 16 == 0) std::this_thread::sleep_for(std::chrono::microseconds(s));
            f();
            executed.fetch_add(1, std::memory_order_release);
        }
    }
};

// every sample of block k on channel c is k * 4 + c: the GPU side can tell a complete, unmixed block and its place in the stream
static void fillBlock(std::vector<float> &store, uint32_t channels, uint32_t n, uint64_t k, const float **ptrs)
{
    store.resize(size_t(channels) * n);
    for (uint32_t c = 0; c < channels; ++c) {
        for (uint32_t i = 0; i < n; ++i) store[size_t(c) * n + i] = float((k % 1000003u) * 4 + c);
        ptrs[c] = store.data() + size_t(c) * n;
    }
}

// ---- scenario A: the batched handles (Oscilloscope / Vectorscope) ------------------------------------------------------------------------
struct BatchedHandle {
    static constexpr uint32_t kChannels = 2, kMaxBlock = 64;
    std::mutex mu;                                       // sgz_scope::mu: configure / set_option / flush lock it, push only tries it
    BatchCore batchCore;
    Backlog bl;
    MockEvent ev[BatchCore::kSlots];
    bool used[BatchCore::kSlots] = {};
    std::atomic<bool> defer{false}, park{false};
    MockStream gpu;
    // what the "ingest kernel" has seen
    std::atomic<uint64_t> consumedBlocks{0};
    uint64_t nextExpected = 0;                           // executor thread only
    std::atomic<uint64_t> busy{0};

    BatchedHandle()
    {
        batchCore.channels = kChannels;
        batchCore.slotSamples = 256;
        batchCore.h = static_cast<float *>(std::malloc(size_t(BatchCore::kSlots) * kChannels * batchCore.slotSamples * sizeof(float)));
        CHECK(bl.init(backlogFloats(kChannels, 2048.0, kMaxBlock)));
    }
    ~BatchedHandle() { gpu.synchronize(); std::free(batchCore.h); bl.release(); }

    // -- the adapter rt_lockfree.hpp's protocol asks for (rt_common.hpp BatchRing + scopeSubmit in the product)
    BatchCore &batch() { return batchCore; }
    Backlog &backlog() { return bl; }
    bool deferSubmit() { return defer.load(std::memory_order_relaxed); }
    bool gpuIdle()
    {
        if (batchCore.seq == 0) return true;
        const int slot = int((batchCore.seq - 1) % BatchCore::kSlots);
        return !used[slot] || ev[slot].done();
    }
    sgz_status slotReady()
    {
        const int slot = int(batchCore.seq % BatchCore::kSlots);
        if (!used[slot]) return SGZ_OK;
        if (!ev[slot].done()) return SGZ_BUSY;
        used[slot] = false;
        return SGZ_OK;
    }
    void waitGpu() { gpu.synchronize(); }
    sgz_status submit()
    {
        CHECK(batchCore.count > 0);
        const int slot = int(batchCore.seq % BatchCore::kSlots);
        const float *base = batchCore.slotBase();
        const uint32_t count = batchCore.count;
        uint32_t off[BatchCore::kMaxBlocks], len[BatchCore::kMaxBlocks];
        for (uint32_t b = 0; b < count; ++b) { off[b] = batchCore.off[b]; len[b] = batchCore.len[b]; }
        std::vector<uint32_t> o(off, off + count), l(len, len + count);
        gpu.enqueue([this, base, o, l] {                 // the ingest kernel: reads the slot's plain memory
            for (size_t b = 0; b < o.size(); ++b) {
                const float *blk = base + o[b];
                const float want0 = float((nextExpected % 1000003u) * 4);
                for (uint32_t c = 0; c < kChannels; ++c)
                    for (uint32_t i = 0; i < l[b]; ++i) CHECK(blk[size_t(c) * l[b] + i] == want0 + float(c));
                ++nextExpected;
            }
            consumedBlocks.fetch_add(o.size(), std::memory_order_release);
        });
        gpu.record(ev[slot]);
        used[slot] = true;
        batchCore.committed();
        return SGZ_OK;
    }

    // -- the C ABI's shape (scope_stream.hip sgz_scope_push / _flush / _configure / _set_option and a reader)
    sgz_status push(const float *const *planar, uint32_t n)
    {
        std::unique_lock<std::mutex> lk(mu, std::try_to_lock);
        if (!lk.owns_lock()) { busy++; return SGZ_BUSY; }
        const sgz_status st = batchPush(*this, planar, kChannels, n, park.load(std::memory_order_relaxed));
        if (st == SGZ_BUSY) busy++;
        return st;
    }
    uint64_t read()                                      // any render-thread reader: flush on read, then its own work behind it, then wait
    {
        CHECK(batchSync(*this) == SGZ_OK);
        gpu.synchronize();
        return consumedBlocks.load(std::memory_order_acquire);
    }
    void flush()
    {
        std::lock_guard<std::mutex> lk(mu);
        CHECK(batchFlushAll(*this) == SGZ_OK);
    }
    void configure()                                     // the audio already taken goes through the old configuration, then everything is reset
    {
        std::lock_guard<std::mutex> lk(mu);
        CHECK(batchFlushAll(*this) == SGZ_OK);
        gpu.synchronize();
        bl.clear();
        batchCore.seq = 0; batchCore.count = batchCore.samples = 0;
        for (auto &u : used) u = false;
    }
    void setOption(bool deferValue, bool parkValue)
    {
        std::lock_guard<std::mutex> lk(mu);
        defer.store(deferValue, std::memory_order_relaxed);
        park.store(parkValue, std::memory_order_relaxed);
    }
};

static void scenarioBatched(uint64_t blocks)
{
    BatchedHandle h;
    std::mutex consumerSide;                              // the host's own serialisation of configure against the readers (same thread in the plugin)
    std::atomic<bool> producerDone{false};
    std::atomic<uint64_t> pushed{0};

    std::thread producer([&] {
        std::minstd_rand rng(1);
        std::vector<float> store;
        const float *ptrs[BatchedHandle::kChannels];
        for (uint64_t k = 0; k < blocks; ++k) {
            const uint32_t n = 1 + rng() % BatchedHandle::kMaxBlock;
            fillBlock(store, BatchedHandle::kChannels, n, k, ptrs);
            while (h.push(ptrs, n) != SGZ_OK) { opTick(); std::this_thread::yield(); }      // refused (FIFO full / reconfiguration): offered again
            opTick();
            pushed.store(k + 1, std::memory_order_release);
            if (k % 50000 == 0) h.gpu.slowness.store(k % 100000 ? 20 : 0, std::memory_order_relaxed);
        }
        producerDone.store(true, std::memory_order_release);
    });
    std::thread control([&] {
        std::minstd_rand rng(2);
        uint32_t i = 0;
        while (!producerDone.load(std::memory_order_acquire)) {
            ++i;
            h.setOption((i / 3) % 2 != 0, (i / 5) % 3 == 0);
            opTick();
            if (i % 7 == 0) {
                std::lock_guard<std::mutex> lk(consumerSide);
                h.configure();
                opTick();
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50 + rng() % 200));
        }
        h.setOption(false, false);
    });
    std::thread consumer([&] {
        uint64_t last = 0;
        while (!producerDone.load(std::memory_order_acquire)) {
            std::lock_guard<std::mutex> lk(consumerSide);
            const uint64_t before = pushed.load(std::memory_order_acquire);
            const uint64_t seen = h.read();
            opTick();
            CHECK(seen >= last);
            // flush on read: every block whose push had returned before this read began is on the GPU side now -- parked ones too
            CHECK(seen >= before);
            last = seen;
        }
    });
    producer.join();
    control.join();
    consumer.join();
    // a stopped transport: blocks parked while no further push comes must reach the GPU with the next read
    {
        std::vector<float> store;
        const float *ptrs[BatchedHandle::kChannels];
        h.setOption(true, true);
        for (uint64_t k = blocks; k < blocks + 5; ++k) {
            fillBlock(store, BatchedHandle::kChannels, 17, k, ptrs);
            CHECK(h.push(ptrs, 17) == SGZ_OK);
        }
        CHECK(h.read() == blocks + 5);
        h.flush();
    }
    CHECK(h.consumedBlocks.load() == blocks + 5);
    std::printf("batched handle: %llu blocks, %llu waited in the host FIFO, %llu pushes refused (FIFO full / reconfiguration)\n",
                static_cast<unsigned long long>(blocks + 5), static_cast<unsigned long long>(h.bl.deferred), static_cast<unsigned long long>(h.busy.load()));
}

// ---- scenario B: the spectrum handle (push lock + FIFO; column queue; line-results triple buffer) ----------------------------------------
struct SpectrumHandle {
    static constexpr uint32_t kChannels = 2, kMaxBlock = 64;
    static constexpr int kStageSlots = 8, kQueueDepth = 10, kLineSlots = 3, kP = 32;
    std::mutex cfgMu;                                    // sgz_spectrum::cfgMu: the reader's right over the FIFO
    Backlog bl;
    MockStream gpu;
    // staging ring (rt_common.hpp StageRing, reduced to its slot arithmetic)
    float *stage = nullptr;
    MockEvent stageEv[kStageSlots];
    bool stageUsed[kStageSlots] = {};
    uint64_t stageSeq = 0;
    uint64_t nextExpected = 0;                           // executor thread only
    // outputs
    ColumnQueue<kQueueDepth> colQ;
    uint32_t cols[kQueueDepth][kP];                      // PLAIN memory: the slot hand-over is what ThreadSanitizer checks
    MockEvent colEv[kQueueDepth];
    LineSeqlock<kLineSlots> lineSeq;
    std::atomic<uint32_t> lines[kLineSlots][kP];         // relaxed atomics: see the file comment
    MockEvent lineEv[kLineSlots];
    std::atomic<uint64_t> dropped{0}, busy{0}, frames{0};

    SpectrumHandle()
    {
        stage = static_cast<float *>(std::malloc(size_t(kStageSlots) * kChannels * kMaxBlock * sizeof(float)));
        CHECK(bl.init(backlogFloats(kChannels, 2048.0, kMaxBlock)));
        for (auto &s : lines) for (auto &w : s) w.store(0, std::memory_order_relaxed);
    }
    ~SpectrumHandle() { gpu.synchronize(); std::free(stage); bl.release(); }

    // realtime.hip spectrumPushNow + emitFrames: one block into a staging slot, behind it the kernels that consume it and one "frame"
    sgz_status pushNow(const float *const *blk, uint32_t n)
    {
        const int slot = int(stageSeq % kStageSlots);
        if (stageUsed[slot] && !stageEv[slot].done()) return SGZ_BUSY;
        float *hs = stage + size_t(slot) * kChannels * kMaxBlock;
        for (uint32_t c = 0; c < kChannels; ++c) std::memcpy(hs + size_t(c) * n, blk[c], size_t(n) * sizeof(float));
        gpu.enqueue([this, hs, n] {
            const float want0 = float((nextExpected % 1000003u) * 4);
            for (uint32_t c = 0; c < kChannels; ++c)
                for (uint32_t i = 0; i < n; ++i) CHECK(hs[size_t(c) * n + i] == want0 + float(c));
            ++nextExpected;
        });
        gpu.record(stageEv[slot]);
        stageUsed[slot] = true;
        ++stageSeq;
        const uint64_t frame = frames.fetch_add(1, std::memory_order_relaxed) + 1;
        // line results of the newest frame -> the host's triple buffer
        {
            const uint64_t n2 = lineSeq.begin();
            const int ls = lineSeq.slotOf(n2);
            gpu.enqueue([this, ls, n2] { for (auto &w : lines[ls]) w.store(uint32_t(n2), std::memory_order_relaxed); });
            gpu.record(lineEv[ls]);
            lineSeq.publish(n2);
        }
        // the frame's column into the queue (dropped when the consumer is ten columns behind: SpectrumDSP.cpp:185-186)
        int cs = 0;
        if (!colQ.producerSlot(&cs)) { dropped++; return SGZ_OK; }
        gpu.enqueue([this, cs, frame] { for (auto &w : cols[cs]) w = uint32_t(frame); });
        gpu.record(colEv[cs]);
        colQ.producerPublish();
        return SGZ_OK;
    }
    sgz_status push(const float *const *planar, uint32_t n)
    {
        std::unique_lock<std::mutex> lk(cfgMu, std::try_to_lock);
        if (!lk.owns_lock()) { busy++; return SGZ_BUSY; }
        auto now = [&](const float *const *blk, uint32_t, uint32_t m) -> sgz_status { return pushNow(blk, m); };
        const sgz_status st = pushThroughBacklog(bl, planar, kChannels, n, now);
        if (st == SGZ_BUSY) busy++;
        return st;
    }
    void flush()                                         // realtime.hip sgz_spectrum_flush
    {
        std::lock_guard<std::mutex> lk(cfgMu);
        const float *ptrs[64];
        while (bl.count.load(std::memory_order_acquire)) {
            const Backlog::Entry e = bl.front();
            for (uint32_t c = 0; c < e.channels; ++c) ptrs[c] = bl.buf + e.off + size_t(c) * e.n;
            const sgz_status st = pushNow(ptrs, e.n);
            if (st == SGZ_BUSY) { gpu.synchronize(); continue; }
            bl.pop();
        }
    }
    // sgz_spectrum_pop_column: returns the frame number of the column, 0 when none is ready
    uint32_t popColumn()
    {
        const uint64_t head = colQ.consumerHead();
        if (!colQ.consumerHas(head)) return 0;
        const int slot = int(head % kQueueDepth);
        if (!colEv[slot].done()) return 0;
        uint32_t out[kP];
        for (int i = 0; i < kP; ++i) out[i] = cols[slot][i];        // (a loop, not memcpy: gcc expands a fixed-size memcpy inline WITHOUT ThreadSanitizer instrumentation)
        colQ.consumerRelease(head + 1);
        for (int i = 1; i < kP; ++i) CHECK(out[i] == out[0]);
        return out[0];
    }
    // sgz_spectrum_line_results: the copy number read (0: nothing published yet), never a torn one
    uint64_t lineResults()
    {
        auto landed = [&](int slot) { return lineEv[slot].done() && lineEv[slot].everRecorded(); };
        for (int attempt = 0; attempt < 8; ++attempt) {
            bool none = false;
            const uint64_t n = lineSeq.newest(landed, &none);
            if (n == 0) { if (none) return 0; continue; }
            uint32_t out[kP];
            for (int i = 0; i < kP; ++i) out[i] = lines[lineSeq.slotOf(n)][i].load(std::memory_order_relaxed);
            if (!lineSeq.stillValid(n)) continue;
            for (int i = 0; i < kP; ++i) CHECK(out[i] == uint32_t(n));
            return n;
        }
        gpu.synchronize();                                // (the product falls back to the device copy behind the producer's work)
        return ~uint64_t(0);
    }
};

static void scenarioSpectrum(uint64_t blocks)
{
    SpectrumHandle h;
    std::atomic<bool> producerDone{false};
    std::thread producer([&] {
        std::minstd_rand rng(3);
        std::vector<float> store;
        const float *ptrs[SpectrumHandle::kChannels];
        for (uint64_t k = 0; k < blocks; ++k) {
            const uint32_t n = 1 + rng() % SpectrumHandle::kMaxBlock;
            fillBlock(store, SpectrumHandle::kChannels, n, k, ptrs);
            while (h.push(ptrs, n) != SGZ_OK) { opTick(); std::this_thread::yield(); }
            opTick();
            if (k % 40000 == 0) h.gpu.slowness.store(k % 80000 ? 15 : 0, std::memory_order_relaxed);
        }
        producerDone.store(true, std::memory_order_release);
    });
    std::thread control([&] {                            // the consumer side's flush (takes the push lock: pushes meanwhile are refused and re-offered)
        while (!producerDone.load(std::memory_order_acquire)) {
            h.flush();
            opTick();
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
    });
    std::thread consumer([&] {
        uint32_t lastCol = 0;
        uint64_t lastLine = 0, fallbacks = 0;
        while (!producerDone.load(std::memory_order_acquire) || h.colQ.consumerHas(h.colQ.consumerHead())) {
            const uint32_t c = h.popColumn();
            opTick();
            if (c) { CHECK(c > lastCol); lastCol = c; }   // in order, none twice (a full queue drops frames: gaps are allowed)
            const uint64_t n = h.lineResults();
            opTick();
            if (n == ~uint64_t(0)) { ++fallbacks; continue; }
            CHECK(n >= lastLine);
            lastLine = n;
            if (!c && producerDone.load(std::memory_order_acquire)) h.gpu.synchronize();
        }
        (void)fallbacks;
    });
    producer.join();
    control.join();
    consumer.join();
    h.flush();
    h.gpu.synchronize();
    CHECK(h.nextExpected == blocks);                     // every block reached the GPU side, in order, exactly once
    CHECK(h.lineResults() == blocks);                    // the newest copy, whole
    std::printf("spectrum handle: %llu blocks, %llu waited in the host FIFO, %llu pushes refused, %llu columns dropped by a full queue\n",
                static_cast<unsigned long long>(blocks), static_cast<unsigned long long>(h.bl.deferred), static_cast<unsigned long long>(h.busy.load()),
                static_cast<unsigned long long>(h.dropped.load()));
}

int main(int argc, char **argv)
{
    const uint64_t blocks = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 250000;
    scenarioBatched(blocks);
    scenarioSpectrum(blocks);
    const uint64_t ops = g_ops.load();
    std::printf("tsan harness ok: %llu operations (push / read / flush / configure / set_option / pop_column / line_results calls), %llu blocks per scenario\n",
                static_cast<unsigned long long>(ops), static_cast<unsigned long long>(blocks));
    return 0;
}
