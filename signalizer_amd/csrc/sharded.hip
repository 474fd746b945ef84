// sharded.hip -- multi-GPU time-chunk render behind the C ABI: one process per GPU, rank r owns a contiguous chunk
// of the stream and every frame whose first sample lies in it.  The same protocol as signalizer_amd/sharding.py, on the host's own
// RCCL communicator, so that a C++ host needs neither torch nor Python:
//   A1  halo: ncclSend of this rank's leading samples to rank - 1 / ncclRecv of rank + 1's (exactly the samples the last frames
//       reach into the next chunk: W - hop when hop divides the chunk), one grouped pair -- each transfer rides one xGMI link.
//       It runs on a stream of its own WHILE K_A transforms the frames that lie inside the chunk (all but the last
//       ceil((W - hop) / hop) = 3 at 75 % overlap); only those last frames wait for it;
//   K_A over the local frames (two launches: inside the chunk / reaching into the halo); K_B scan from a zero carry-in -> end state;
//   A2  ncclAllGather of the end states (pairs x graphs x P x 2 floats per rank) -> exact carry fold (decayFoldKernel);
//   K_B emit with the carry folded into the kept aggregates.
// The three collectives go through a small table of functions (sgz_transport, sgz.h): RCCL's by default, the caller's own otherwise --
// which is also how the tests run this very code with 2-4 ranks sharing one GPU (RCCL refuses that; a host-memory shim does not).
// RCCL is bound at run time (dlopen): libsgz.so has no link-time dependency on it, and inside a PyTorch process the RCCL that
// torch already loaded is the one used.  The real-time per-block path stays single-GPU ("replicas only").
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "runtime.hpp"

using namespace sgz;

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclFloat = 7;          // ncclFloat32
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rcclOnce;

Rccl *rccl()
{
    std::call_once(g_rcclOnce, [] {
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);            // PyTorch's bundled copy, if this process has it
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
        if (!h) h = dlopen("librccl.so", RTLD_NOW);
        if (!h) return;
        Rccl r;
        r.lib = h;
#define SGZ_SYM(field, name) *reinterpret_cast<void **>(&r.field) = dlsym(h, name)
        SGZ_SYM(GetUniqueId, "ncclGetUniqueId"); SGZ_SYM(CommInitRank, "ncclCommInitRank"); SGZ_SYM(CommDestroy, "ncclCommDestroy"); SGZ_SYM(CommAbort, "ncclCommAbort");
        SGZ_SYM(AllGather, "ncclAllGather"); SGZ_SYM(Send, "ncclSend"); SGZ_SYM(Recv, "ncclRecv");
        SGZ_SYM(GroupStart, "ncclGroupStart"); SGZ_SYM(GroupEnd, "ncclGroupEnd"); SGZ_SYM(GetErrorString, "ncclGetErrorString");
#undef SGZ_SYM
        if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd) g_rccl = r;
    });
    return g_rccl.lib ? &g_rccl : nullptr;
}

sgz_status ncclFail(ncclResult_t e, const char *what)
{
    const Rccl *r = rccl();
    return fail(SGZ_EHIP, std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}
#define SGZ_NCCL(call) do { ncclResult_t _e = (call); if (_e != 0) return ncclFail(_e, #call); } while (0)

// partition arithmetic (signalizer_amd/sharding.py ShardPlan; tests/test_sharding.py pins it)
struct Shard {
    uint64_t S, W, hop; uint32_t world;
    uint64_t totalFrames() const { const uint64_t total = S * world; return total < W ? 0 : (total - W) / hop + 1; }
    uint64_t firstFrame(uint32_t r) const { const uint64_t f = (uint64_t(r) * S + hop - 1) / hop; const uint64_t t = totalFrames(); return f < t ? f : t; }
    uint64_t framesOf(uint32_t r) const { return (r + 1 < world ? firstFrame(r + 1) : totalFrames()) - firstFrame(r); }
    uint64_t localOffset(uint32_t r) const { return firstFrame(r) * hop - uint64_t(r) * S; }
    uint64_t halo(uint32_t r) const
    {
        const uint64_t f = framesOf(r);
        if (!f) return 0;
        const uint64_t lastEnd = (firstFrame(r) + f - 1) * hop + W, own = uint64_t(r + 1) * S;
        return lastEnd > own ? lastEnd - own : 0;
    }
};

// ---- A transport for hosts that drive several GPUs from ONE process (a thread per rank): hipMemcpyPeerAsync over xGMI, no RCCL.
// Every transfer is a peer copy enqueued by the RECEIVER on its own stream behind an event the sender recorded on its stream; the
// sender's stream in turn waits for the event the receiver records behind the copy, so that -- as with ncclSend / ncclAllGather -- a
// buffer belongs to the transfer until the call is complete ON THE STREAM, whichever side is faster.  The host side only hands
// pointers and events across (mutex + condition variable): no thread ever waits for a GPU, only for its peer to have ENQUEUED.
struct PeerSlot {
    const float *ptr = nullptr; size_t count = 0;
    uint64_t posted = 0, taken = 0;                 // messages the sender has posted / the receiver has enqueued the copy of
    hipEvent_t ready = nullptr, done = nullptr;     // recorded by the sender behind its data / by the receiver behind its copy
};
}  // namespace
struct sgz_peer_group {
    uint32_t world = 0;
    std::vector<int> device;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<PeerSlot> p2p;                      // [src * world + dst]: send / recv
    std::vector<PeerSlot> ag;                       // [src * world + dst]: all-gather, src's block as read by dst
    std::vector<uint64_t> agRound;                  // [rank]: all-gathers this rank has entered
    std::vector<uint64_t> pendingSend;              // [src * world + dst]: message whose `done` the sender's stream still has to wait for (0 = none)
    bool aborted = false;
};
struct PeerCtx { sgz_peer_group *g; uint32_t rank; };
namespace {
constexpr int kPeerErr = 1000;
int peerWaitTaken(sgz_peer_group *g, PeerSlot &sl, uint64_t msg, hipStream_t stream)
{
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return sl.taken >= msg || g->aborted; });
    if (g->aborted) return kPeerErr;
    lk.unlock();
    return hipStreamWaitEvent(stream, sl.done, 0) == hipSuccess ? 0 : kPeerErr + 1;
}
int peerSend(void *ctx, const float *d_buf, size_t count, uint32_t peer, void *stream)
{
    PeerCtx *c = static_cast<PeerCtx *>(ctx);
    sgz_peer_group *g = c->g;
    if (peer >= g->world || peer == c->rank) return kPeerErr + 2;
    PeerSlot &sl = g->p2p[size_t(c->rank) * g->world + peer];
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // the previous message to this peer: its buffer is the same one more often than not
    if (g->pendingSend[size_t(c->rank) * g->world + peer])
        if (int e = peerWaitTaken(g, sl, g->pendingSend[size_t(c->rank) * g->world + peer], s); e) return e;
    if (hipEventRecord(sl.ready, s) != hipSuccess) return kPeerErr + 3;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        sl.ptr = d_buf; sl.count = count; sl.posted++;
        g->pendingSend[size_t(c->rank) * g->world + peer] = sl.posted;
    }
    g->cv.notify_all();
    return 0;
}
int peerRecv(void *ctx, float *d_buf, size_t count, uint32_t peer, void *stream)
{
    PeerCtx *c = static_cast<PeerCtx *>(ctx);
    sgz_peer_group *g = c->g;
    if (peer >= g->world || peer == c->rank) return kPeerErr + 2;
    PeerSlot &sl = g->p2p[size_t(peer) * g->world + c->rank];
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const float *src; size_t n;
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->cv.wait(lk, [&] { return sl.posted > sl.taken || g->aborted; });
        if (g->aborted) return kPeerErr;
        src = sl.ptr; n = sl.count;
    }
    if (n != count) return kPeerErr + 4;
    if (hipStreamWaitEvent(s, sl.ready, 0) != hipSuccess) return kPeerErr + 5;
    if (hipMemcpyPeerAsync(d_buf, g->device[c->rank], src, g->device[peer], count * sizeof(float), s) != hipSuccess) return kPeerErr + 6;
    if (hipEventRecord(sl.done, s) != hipSuccess) return kPeerErr + 7;
    { std::lock_guard<std::mutex> lk(g->mu); sl.taken++; }
    g->cv.notify_all();
    return 0;
}
// group_end: the sends of this exchange are complete on the stream once the receivers' copies are
int peerGroupEnd(void *ctx)
{
    (void)ctx;                                       // (nothing to close: a send's completion is awaited by the next send / the all-gather)
    return 0;
}
int peerAllGather(void *ctx, const float *d_send, float *d_recv, size_t count, void *stream)
{
    PeerCtx *c = static_cast<PeerCtx *>(ctx);
    sgz_peer_group *g = c->g;
    const uint32_t W = g->world, r = c->rank;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // outstanding halo sends of this call complete here at the latest (their buffers are the caller's again after the render)
    for (uint32_t q = 0; q < W; ++q)
        if (uint64_t m = g->pendingSend[size_t(r) * W + q]) {
            if (int e = peerWaitTaken(g, g->p2p[size_t(r) * W + q], m, s); e) return e;
            g->pendingSend[size_t(r) * W + q] = 0;
        }
    // my block is ready behind this event; every reader (myself included: a plain device copy) gets its own slot
    PeerSlot &mine = g->ag[size_t(r) * W + r];
    if (hipEventRecord(mine.ready, s) != hipSuccess) return kPeerErr + 3;
    uint64_t round;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        round = ++g->agRound[r];
        for (uint32_t q = 0; q < W; ++q) { PeerSlot &sl = g->ag[size_t(r) * W + q]; sl.ptr = d_send; sl.count = count; sl.posted = round; }
    }
    g->cv.notify_all();
    for (uint32_t q = 0; q < W; ++q) {                // read rank q's block
        PeerSlot &sl = g->ag[size_t(q) * W + r];
        const float *src; size_t n;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return sl.posted >= round || g->aborted; });
            if (g->aborted) return kPeerErr;
            src = sl.ptr; n = sl.count;
        }
        if (n != count) return kPeerErr + 4;
        if (hipStreamWaitEvent(s, g->ag[size_t(q) * W + q].ready, 0) != hipSuccess) return kPeerErr + 5;
        if (hipMemcpyPeerAsync(d_recv + size_t(q) * count, g->device[r], src, g->device[q], count * sizeof(float), s) != hipSuccess) return kPeerErr + 6;
        if (hipEventRecord(sl.done, s) != hipSuccess) return kPeerErr + 7;
        { std::lock_guard<std::mutex> lk(g->mu); sl.taken = round; }
        g->cv.notify_all();
    }
    // like ncclAllGather, the operation is complete on this stream only when everybody has read my block
    for (uint32_t q = 0; q < W; ++q)
        if (q != r)
            if (int e = peerWaitTaken(g, g->ag[size_t(r) * W + q], round, s); e) return e;
    return 0;
}
void peerAbort(void *ctx)
{
    sgz_peer_group *g = static_cast<PeerCtx *>(ctx)->g;
    { std::lock_guard<std::mutex> lk(g->mu); g->aborted = true; }
    g->cv.notify_all();
}

}  // namespace

struct sgz_plan { Plan impl; };

extern "C" {

sgz_status sgz_comm_unique_id(uint8_t out[128])
{
    Rccl *r = rccl();
    if (!r) return fail(SGZ_EHIP, "RCCL (librccl.so) not found");
    if (!out) return fail(SGZ_EINVAL, "null argument");
    ncclUniqueId id;
    SGZ_NCCL(r->GetUniqueId(&id));
    std::memcpy(out, id.internal, 128);
    return SGZ_OK;
}

sgz_status sgz_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, void **comm)
{
    Rccl *r = rccl();
    if (!r) return fail(SGZ_EHIP, "RCCL (librccl.so) not found");
    if (!id || !comm || rank >= world) return fail(SGZ_EINVAL, "bad argument");
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, 128);
    ncclComm_t c = nullptr;
    SGZ_NCCL(r->CommInitRank(&c, int(world), uid, int(rank)));
    *comm = c;
    return SGZ_OK;
}

void sgz_comm_destroy(void *comm)
{
    Rccl *r = rccl();
    if (r && comm) (void)r->CommDestroy(static_cast<ncclComm_t>(comm));
}

sgz_status sgz_shard_layout(const sgz_plan *plan, uint32_t rank, uint32_t world, size_t chunk_samples, uint64_t *local_frames,
                            uint64_t *first_frame, uint64_t *halo_in, uint64_t *halo_out)
{
    if (!plan || rank >= world || world == 0) return fail(SGZ_EINVAL, "bad argument");
    const Plan &p = plan->impl;
    if (isResonator(p)) {
        // RSNT: a frame per `hop` samples, no window history -- whole hops per chunk, no halo
        if (chunk_samples == 0 || chunk_samples % p.cfg.hop) return fail(SGZ_EINVAL, "RSNT: a chunk is a whole number of hops");
        if (local_frames) *local_frames = chunk_samples / p.cfg.hop;
        if (first_frame) *first_frame = uint64_t(rank) * (chunk_samples / p.cfg.hop);
        if (halo_in) *halo_in = 0;
        if (halo_out) *halo_out = 0;
        return SGZ_OK;
    }
    if (chunk_samples < p.W) return fail(SGZ_EINVAL, "a chunk must hold at least one window");
    const Shard sh{chunk_samples, p.W, p.cfg.hop, world};
    if (local_frames) *local_frames = sh.framesOf(rank);
    if (first_frame) *first_frame = sh.firstFrame(rank);
    if (halo_in) *halo_in = sh.halo(rank);
    if (halo_out) *halo_out = rank ? sh.halo(rank - 1) : 0;
    return SGZ_OK;
}

// ---- RCCL as an sgz_transport
static int rcclSend(void *ctx, const float *d, size_t n, uint32_t peer, void *stream)
{
    return rccl()->Send(d, n, kNcclFloat, int(peer), static_cast<ncclComm_t>(ctx), reinterpret_cast<hipStream_t>(stream));
}
static int rcclRecv(void *ctx, float *d, size_t n, uint32_t peer, void *stream)
{
    return rccl()->Recv(d, n, kNcclFloat, int(peer), static_cast<ncclComm_t>(ctx), reinterpret_cast<hipStream_t>(stream));
}
static int rcclAllGather(void *ctx, const float *d_send, float *d_recv, size_t nPerRank, void *stream)
{
    return rccl()->AllGather(d_send, d_recv, nPerRank, kNcclFloat, static_cast<ncclComm_t>(ctx), reinterpret_cast<hipStream_t>(stream));
}
static int rcclGroupBegin(void *) { return rccl()->GroupStart(); }
static int rcclGroupEnd(void *) { return rccl()->GroupEnd(); }
static void rcclAbort(void *ctx) { if (rccl()->CommAbort) (void)rccl()->CommAbort(static_cast<ncclComm_t>(ctx)); }

// ---- the peer-copy transport's lifetime (sgz.h)
sgz_status sgz_peer_group_create(uint32_t world, const int *devices, sgz_peer_group **out)
{
    if (!devices || !out || world == 0 || world > 64) return fail(SGZ_EINVAL, "bad argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(SGZ_EHIP, "no HIP device visible");
    int keep = 0;
    (void)hipGetDevice(&keep);
    sgz_peer_group *g = new (std::nothrow) sgz_peer_group();
    if (!g) return fail(SGZ_ENOMEM, "out of memory");
    g->world = world;
    g->device.assign(devices, devices + world);
    g->p2p.resize(size_t(world) * world); g->ag.resize(size_t(world) * world);
    g->agRound.assign(world, 0); g->pendingSend.assign(size_t(world) * world, 0);
    sgz_status st = SGZ_OK;
    for (uint32_t a = 0; a < world && st == SGZ_OK; ++a) {
        if (devices[a] < 0 || devices[a] >= count) { st = fail(SGZ_EINVAL, "device index out of range"); break; }
        for (uint32_t b = 0; b < world && st == SGZ_OK; ++b) {
            // `ready` is recorded on a stream of the source's device, `done` on one of the reader's
            for (int which = 0; which < 2 && st == SGZ_OK; ++which) {
                std::vector<PeerSlot> &v = which ? g->ag : g->p2p;
                PeerSlot &sl = v[size_t(a) * world + b];
                if (hipSetDevice(devices[a]) != hipSuccess || hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming) != hipSuccess ||
                    hipSetDevice(devices[b]) != hipSuccess || hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess)
                    st = fail(SGZ_EHIP, "event creation failed");
            }
            if (st == SGZ_OK && devices[a] != devices[b]) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, devices[b], devices[a]);
                if (can) { (void)hipSetDevice(devices[b]); const hipError_t e = hipDeviceEnablePeerAccess(devices[a], 0); if (e != hipSuccess) (void)hipGetLastError(); }   // (already enabled is fine; without access the copies are staged by the runtime)
            }
        }
    }
    (void)hipSetDevice(keep);
    if (st != SGZ_OK) { sgz_peer_group_destroy(g); return st; }
    *out = g;
    return SGZ_OK;
}

void sgz_peer_group_destroy(sgz_peer_group *g)
{
    if (!g) return;
    for (std::vector<PeerSlot> *v : {&g->p2p, &g->ag})
        for (PeerSlot &sl : *v) { if (sl.ready) (void)hipEventDestroy(sl.ready); if (sl.done) (void)hipEventDestroy(sl.done); }
    delete g;
}

sgz_status sgz_peer_transport(sgz_peer_group *g, uint32_t rank, sgz_transport *out, void **ctx_storage)
{
    if (!g || !out || !ctx_storage || rank >= g->world) return fail(SGZ_EINVAL, "bad argument");
    PeerCtx *c = new (std::nothrow) PeerCtx{g, rank};
    if (!c) return fail(SGZ_ENOMEM, "out of memory");
    *ctx_storage = c;
    *out = sgz_transport{c, peerSend, peerRecv, peerAllGather, nullptr, peerGroupEnd, peerAbort};
    return SGZ_OK;
}
void sgz_peer_transport_release(void *ctx_storage) { delete static_cast<PeerCtx *>(ctx_storage); }

sgz_status sgz_spectrogram_render_sharded(sgz_plan *plan, void *nccl_comm, uint32_t rank, uint32_t world, float *d_chunk,
                                          size_t channel_stride, size_t chunk_samples, uint8_t *d_rgba, uint64_t *local_frames,
                                          void *stream)
{
    if (!nccl_comm || !rccl()) return fail(SGZ_EHIP, "no RCCL communicator");
    const sgz_transport t{nccl_comm, rcclSend, rcclRecv, rcclAllGather, rcclGroupBegin, rcclGroupEnd, rcclAbort};
    return sgz_spectrogram_render_sharded_on(plan, &t, rank, world, d_chunk, channel_stride, chunk_samples, d_rgba, local_frames, stream);
}

// RSNT across ranks.  The resonator recurrence s[n] = c s[n-1] + x[n] is linear, so a rank renders its chunk FROM REST and is then
// short, in every frame f, of c^((f + 1) hop) x (the state that entered its chunk).  That state is a fold of the ranks' end states --
// one all-gather of [C][2][V][P] complex values (196 KB per rank at cfg2 sizes) -- evaluated in fp64 (launchResonatorFold), added to the
// chained states (launchResonatorCarry), and only then do the window kernel and K_B run; K_B's own carry travels as for the FFT
// plans.  No halo: an RSNT frame consumes exactly its `hop` samples.  Against a single-device render the frames differ by the
// roundings of that one extra fp64 -> fp32 addition per frame: the same bar as the chained frames of a single device
// (tests/test_gpu_resonator.py), not bit-identity.
static sgz_status renderShardedResonator(Plan &p, const sgz_transport *t, uint32_t rank, uint32_t world, float *d_chunk, size_t channel_stride,
                                         size_t chunk_samples, uint8_t *d_rgba, uint64_t *local_frames, hipStream_t s)
{
    if (chunk_samples == 0 || chunk_samples % p.cfg.hop) return fail(SGZ_EINVAL, "RSNT: a chunk is a whole number of hops");
    if (channel_stride < chunk_samples) return fail(SGZ_EINVAL, "channel_stride shorter than the chunk");
    const long frames = long(chunk_samples / p.cfg.hop);
    if (local_frames) *local_frames = uint64_t(frames);
    auto bail = [&](sgz_status st) { if (world > 1 && t->abort) t->abort(t->ctx); return st; };
    auto coll = [&](int e, const char *what) { return e == 0 ? SGZ_OK : bail(fail(SGZ_EHIP, std::string(what) + " failed (transport error " + std::to_string(e) + ")")); };
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    const size_t resN = size_t(p.C) * 2 * size_t(p.resV) * p.P * 2;              // the resonators' state, floats
    if (sgz_status bs = checkResonatorShardBound(p, frames); bs != SGZ_OK) return bail(bs);      // before anything is allocated or exchanged
    // work buffer: [decay end state][decay carry][world x decay end states][resonator carry][world x resonator end states]
    sgz_status st = ensureCap(&p.d_shard, &p.shardCap, stateN * (2 + world) + resN * (1 + world));
    if (st != SGZ_OK) return bail(st);
    float *d_end = p.d_shard, *d_carry = d_end + stateN, *d_all = d_carry + stateN, *d_resCarry = d_all + stateN * world, *d_resAll = d_resCarry + resN;
    if ((st = ensureCap(&p.d_mapped, &p.mappedCap, size_t(frames) * p.C * p.sides * p.P)) != SGZ_OK) return bail(st);
    if ((st = runResonatorFromRest(p, d_chunk, channel_stride, frames, p.d_mapped, s)) != SGZ_OK) return bail(st);
    if ((st = coll(t->allgather(t->ctx, p.d_resState, d_resAll, resN, s), "resonator end-state all-gather")) != SGZ_OK) return st;
    long long fr[64];
    for (uint32_t q = 0; q < world; ++q) fr[q] = frames;
    if ((st = runResonatorJoin(p, frames, p.d_mapped, d_resAll, fr, world, rank, d_resCarry, s)) != SGZ_OK) return bail(st);
    // the decay filters: zero-carry scan -> all-gather -> exact fold -> emit, as for the FFT plans
    if (hipError_t e = hipMemsetAsync(d_end, 0, stateN * sizeof(float), s); e != hipSuccess) return bail(hipFail(e, "hipMemsetAsync"));
    if ((st = runDecayColour(p, p.d_mapped, frames, nullptr, nullptr, d_end, s, /*magnitudeOnly=*/true)) != SGZ_OK) return bail(st);
    if ((st = coll(t->allgather(t->ctx, d_end, d_all, stateN, s), "end-state all-gather")) != SGZ_OK) return st;
    const float *carry = nullptr;
    if (rank > 0) {
        SGZ_HIP(launchDecayFold(d_all, fr, world, rank, stateN, p.P, p.scalars, d_carry, s));
        carry = d_carry;
    }
    return runDecayEmitWithCarry(p, p.d_mapped, frames, carry, d_rgba, nullptr, nullptr, s);
}

sgz_status sgz_spectrogram_render_sharded_on(sgz_plan *plan, const sgz_transport *t, uint32_t rank, uint32_t world, float *d_chunk,
                                             size_t channel_stride, size_t chunk_samples, uint8_t *d_rgba, uint64_t *local_frames,
                                             void *stream)
{
    if (!plan || !t || !t->send || !t->recv || !t->allgather || !d_chunk || !d_rgba || rank >= world || world == 0 || world > 64)
        return fail(SGZ_EINVAL, "bad argument");
    Plan &p = plan->impl;
    if (!p.uploaded) { std::string err; sgz_status st = uploadPlan(p, err); if (st != SGZ_OK) return fail(st, err); }
    if (isResonator(p)) return renderShardedResonator(p, t, rank, world, d_chunk, channel_stride, chunk_samples, d_rgba, local_frames, reinterpret_cast<hipStream_t>(stream));
    if (chunk_samples < p.W) return fail(SGZ_EINVAL, "a chunk must hold at least one window");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Shard sh{chunk_samples, p.W, p.cfg.hop, world};
    const uint64_t frames = sh.framesOf(rank), haloIn = sh.halo(rank), haloOut = rank ? sh.halo(rank - 1) : 0;
    if (local_frames) *local_frames = frames;
    if (channel_stride < chunk_samples + haloIn) return fail(SGZ_EINVAL, "channel_stride must leave room for the halo behind the chunk");
    const uint32_t nch = 2 * p.C;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    // a failure on this rank alone, once the peers may be inside a collective: they must fail too, not wait for ever
    auto bail = [&](sgz_status st) { if (world > 1 && t->abort) t->abort(t->ctx); return st; };
    auto coll = [&](int e, const char *what) { return e == 0 ? SGZ_OK : bail(fail(SGZ_EHIP, std::string(what) + " failed (transport error " + std::to_string(e) + ")")); };
    // work buffer: [end state][carry][world x end states][halo send][halo recv]
    const size_t need = stateN * (2 + world) + size_t(nch) * (haloOut + haloIn);
    sgz_status st = ensureCap(&p.d_shard, &p.shardCap, need);
    if (st != SGZ_OK) return bail(st);
    float *d_end = p.d_shard, *d_carry = d_end + stateN, *d_all = d_carry + stateN;
    float *d_send = d_all + stateN * world, *d_recv = d_send + size_t(nch) * haloOut;
    // the halo's own stream and the two events that tie it to the caller's
    if (sgz_status s2 = ensureSecondStream(p); s2 != SGZ_OK) return bail(s2);
    hipStream_t cs = static_cast<hipStream_t>(p.shardStream);
    hipEvent_t evFork = static_cast<hipEvent_t>(p.shardEv[0]), evJoin = static_cast<hipEvent_t>(p.shardEv[1]);
    const bool wantRecv = haloIn && rank + 1 < world;
    // A1 on the halo stream: pack -> send / recv (one group) -> unpack behind the chunk
    const bool exchange = haloOut || wantRecv;
    if (exchange) {
        if (hipError_t e = hipEventRecord(evFork, s); e != hipSuccess) return bail(hipFail(e, "hipEventRecord"));      // (the chunk is ready when the caller's stream gets here)
        if (hipError_t e = hipStreamWaitEvent(cs, evFork, 0); e != hipSuccess) return bail(hipFail(e, "hipStreamWaitEvent"));
        if (haloOut)
            if (hipError_t e = hipMemcpy2DAsync(d_send, haloOut * sizeof(float), d_chunk, channel_stride * sizeof(float), haloOut * sizeof(float),
                                                nch, hipMemcpyDeviceToDevice, cs); e != hipSuccess) return bail(hipFail(e, "hipMemcpy2DAsync (halo pack)"));
        int eb = t->group_begin ? t->group_begin(t->ctx) : 0, es = 0, er = 0;
        if (eb == 0 && haloOut) es = t->send(t->ctx, d_send, size_t(nch) * haloOut, rank - 1, cs);
        if (eb == 0 && es == 0 && wantRecv) er = t->recv(t->ctx, d_recv, size_t(nch) * haloIn, rank + 1, cs);
        const int ee = (eb == 0 && t->group_end) ? t->group_end(t->ctx) : 0;      // (a group that was opened is always closed)
        if ((st = coll(eb ? eb : es ? es : er ? er : ee, "halo exchange")) != SGZ_OK) return st;
        if (wantRecv)
            if (hipError_t e = hipMemcpy2DAsync(d_chunk + chunk_samples, channel_stride * sizeof(float), d_recv, haloIn * sizeof(float),
                                                haloIn * sizeof(float), nch, hipMemcpyDeviceToDevice, cs); e != hipSuccess) return bail(hipFail(e, "hipMemcpy2DAsync (halo unpack)"));
        if (hipError_t e = hipEventRecord(evJoin, cs); e != hipSuccess) return bail(hipFail(e, "hipEventRecord"));
    }
    // K_A: the frames inside the chunk first (they overlap the exchange), then -- behind the halo -- the frames that reach into it
    if (frames) {
        if ((st = ensureCap(&p.d_mapped, &p.mappedCap, size_t(frames) * p.C * p.sides * p.P)) != SGZ_OK) return bail(st);
        const uint64_t off = sh.localOffset(rank);
        uint64_t early = frames;
        if (wantRecv) early = chunk_samples >= off + p.W ? std::min<uint64_t>(frames, (chunk_samples - off - p.W) / p.cfg.hop + 1) : 0;
        const size_t perFrame = size_t(p.C) * p.sides * p.P;
        if (early && (st = runStft(p, d_chunk + off, channel_stride, long(early), p.d_mapped, nullptr, nullptr, s)) != SGZ_OK) return bail(st);
        if (exchange)
            if (hipError_t e = hipStreamWaitEvent(s, evJoin, 0); e != hipSuccess) return bail(hipFail(e, "hipStreamWaitEvent"));
        if (early < frames && (st = runStft(p, d_chunk + off + early * p.cfg.hop, channel_stride, long(frames - early),
                                            p.d_mapped + early * perFrame, nullptr, nullptr, s)) != SGZ_OK) return bail(st);
    } else if (exchange) {
        if (hipError_t e = hipStreamWaitEvent(s, evJoin, 0); e != hipSuccess) return bail(hipFail(e, "hipStreamWaitEvent"));
    }
    if (hipError_t e = hipMemsetAsync(d_end, 0, stateN * sizeof(float), s); e != hipSuccess) return bail(hipFail(e, "hipMemsetAsync"));
    if (frames && (st = runDecayColour(p, p.d_mapped, long(frames), nullptr, nullptr, d_end, s, /*magnitudeOnly=*/true)) != SGZ_OK) return bail(st);
    // A2: end states of every rank, exact fold of the predecessors.  Rank 0 has no predecessor: its emit pass does not read the gathered
    // states, so there the all-gather (which the OTHER ranks need its contribution for) runs on the second stream beside it.  On the
    // ranks behind it the carry stands in front of every emitted frame -- with both graphs' poles near 1 it outlives a rank's whole
    // chunk (0.99^44 = 0.64 at 8 ranks), so there is nothing a speculative emit could keep.
    if (rank == 0 && world > 1 && frames) {
        if (hipError_t e = hipEventRecord(evFork, s); e != hipSuccess) return bail(hipFail(e, "hipEventRecord"));
        if (hipError_t e = hipStreamWaitEvent(cs, evFork, 0); e != hipSuccess) return bail(hipFail(e, "hipStreamWaitEvent"));
        if ((st = coll(t->allgather(t->ctx, d_end, d_all, stateN, cs), "end-state all-gather")) != SGZ_OK) return st;
        if (hipError_t e = hipEventRecord(evJoin, cs); e != hipSuccess) return bail(hipFail(e, "hipEventRecord"));
        st = runDecayEmitWithCarry(p, p.d_mapped, long(frames), nullptr, d_rgba, nullptr, nullptr, s);
        // (the call's work is ordered on `s`: the gather's buffers belong to it until the caller's stream has passed this point)
        if (hipError_t e = hipStreamWaitEvent(s, evJoin, 0); e != hipSuccess) return bail(hipFail(e, "hipStreamWaitEvent"));
        return st == SGZ_OK ? SGZ_OK : bail(st);
    }
    if ((st = coll(t->allgather(t->ctx, d_end, d_all, stateN, s), "end-state all-gather")) != SGZ_OK) return st;
    const float *carry = nullptr;
    if (rank > 0) {
        long long fr[64];
        for (uint32_t q = 0; q < world; ++q) fr[q] = (long long)sh.framesOf(q);
        SGZ_HIP(launchDecayFold(d_all, fr, world, rank, stateN, p.P, p.scalars, d_carry, s));
        carry = d_carry;
    }
    if (frames) return runDecayEmitWithCarry(p, p.d_mapped, long(frames), carry, d_rgba, nullptr, nullptr, s);
    return SGZ_OK;
}

}  // extern "C"
