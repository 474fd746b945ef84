// sharded.hip -- multi-GPU time-chunk render behind the C ABI: one process per GPU, rank r owns a contiguous chunk
// of the stream and every frame whose first sample lies in it.  The same protocol as signalizer_amd/sharding.py, on the host's own
// RCCL communicator, so that a C++ host needs neither torch nor Python:
//   A1  halo: ncclSend of this rank's leading samples to rank - 1 / ncclRecv of rank + 1's (exactly the samples the last frames
//       reach into the next chunk: W - hop when hop divides the chunk), one grouped pair -- each transfer rides one xGMI link;
//   K_A over the local frames; K_B scan from a zero carry-in -> this rank's end state;
//   A2  ncclAllGather of the end states (pairs x graphs x P x 2 floats per rank) -> exact carry fold (decayFoldKernel);
//   K_B emit with the carry folded into the kept aggregates.
// RCCL is bound at run time (dlopen): libsgz.so has no link-time dependency on it, and inside a PyTorch process the RCCL that
// torch already loaded is the one used.  The real-time per-block path stays single-GPU ("replicas only").
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

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
        SGZ_SYM(GetUniqueId, "ncclGetUniqueId"); SGZ_SYM(CommInitRank, "ncclCommInitRank"); SGZ_SYM(CommDestroy, "ncclCommDestroy");
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
    if (chunk_samples < p.W) return fail(SGZ_EINVAL, "a chunk must hold at least one window");
    const Shard sh{chunk_samples, p.W, p.cfg.hop, world};
    if (local_frames) *local_frames = sh.framesOf(rank);
    if (first_frame) *first_frame = sh.firstFrame(rank);
    if (halo_in) *halo_in = sh.halo(rank);
    if (halo_out) *halo_out = rank ? sh.halo(rank - 1) : 0;
    return SGZ_OK;
}

sgz_status sgz_spectrogram_render_sharded(sgz_plan *plan, void *nccl_comm, uint32_t rank, uint32_t world, float *d_chunk,
                                          size_t channel_stride, size_t chunk_samples, uint8_t *d_rgba, uint64_t *local_frames,
                                          void *stream)
{
    if (!plan || !d_chunk || !d_rgba || rank >= world || world == 0 || world > 64) return fail(SGZ_EINVAL, "bad argument");
    Plan &p = plan->impl;
    if (!p.uploaded) { std::string err; sgz_status st = uploadPlan(p, err); if (st != SGZ_OK) return fail(st, err); }
    if (p.cfg.channel_mode == SGZ_CH_PHASE)
        return fail(SGZ_EUNSUPPORTED, "Phase mode: the cancellation smoother is a linear recurrence, there is no exact carry fold");
    if (chunk_samples < p.W) return fail(SGZ_EINVAL, "a chunk must hold at least one window");
    Rccl *r = rccl();
    if (!nccl_comm || !r) return fail(SGZ_EHIP, "no RCCL communicator");
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Shard sh{chunk_samples, p.W, p.cfg.hop, world};
    const uint64_t frames = sh.framesOf(rank), haloIn = sh.halo(rank), haloOut = rank ? sh.halo(rank - 1) : 0;
    if (local_frames) *local_frames = frames;
    if (channel_stride < chunk_samples + haloIn) return fail(SGZ_EINVAL, "channel_stride must leave room for the halo behind the chunk");
    const uint32_t nch = 2 * p.C;
    const size_t stateN = size_t(p.C) * SGZ_NUM_GRAPHS * p.P * 2;
    // work buffer: [end state][carry][world x end states][halo send][halo recv]
    const size_t need = stateN * (2 + world) + size_t(nch) * (haloOut + haloIn);
    sgz_status st = ensureCap(&p.d_shard, &p.shardCap, need);
    if (st != SGZ_OK) return st;
    float *d_end = p.d_shard, *d_carry = d_end + stateN, *d_all = d_carry + stateN;
    float *d_send = d_all + stateN * world, *d_recv = d_send + size_t(nch) * haloOut;
    // A1: neighbour halo
    if (haloOut) SGZ_HIP(hipMemcpy2DAsync(d_send, haloOut * sizeof(float), d_chunk, channel_stride * sizeof(float), haloOut * sizeof(float),
                                          nch, hipMemcpyDeviceToDevice, s));
    if (haloOut || (haloIn && rank + 1 < world)) {
        SGZ_NCCL(r->GroupStart());
        if (haloOut) SGZ_NCCL(r->Send(d_send, size_t(nch) * haloOut, kNcclFloat, int(rank) - 1, comm, s));
        if (haloIn && rank + 1 < world) SGZ_NCCL(r->Recv(d_recv, size_t(nch) * haloIn, kNcclFloat, int(rank) + 1, comm, s));
        SGZ_NCCL(r->GroupEnd());
    }
    if (haloIn && rank + 1 < world)
        SGZ_HIP(hipMemcpy2DAsync(d_chunk + chunk_samples, channel_stride * sizeof(float), d_recv, haloIn * sizeof(float),
                                 haloIn * sizeof(float), nch, hipMemcpyDeviceToDevice, s));
    // K_A + zero-carry scan
    if (frames) {
        st = ensureCap(&p.d_mapped, &p.mappedCap, size_t(frames) * p.C * p.sides * p.P);
        if (st != SGZ_OK) return st;
        st = runStft(p, d_chunk + sh.localOffset(rank), channel_stride, long(frames), p.d_mapped, nullptr, nullptr, s);
        if (st != SGZ_OK) return st;
    }
    SGZ_HIP(hipMemsetAsync(d_end, 0, stateN * sizeof(float), s));
    if (frames && (st = runDecayColour(p, p.d_mapped, long(frames), nullptr, nullptr, d_end, s)) != SGZ_OK) return st;
    // A2: end states of every rank, exact fold of the predecessors
    SGZ_NCCL(r->AllGather(d_end, d_all, stateN, kNcclFloat, comm, s));
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
