// ps_shard.hip -- include/putslam_shard.h: sharding over the GPUs of one node for C / C++ hosts, RCCL underneath.
//
// The exchanges are those of bench.py / putslam_amd/sharding.py: one broadcast of the run's parameter
// block, one gather of 72-byte per-pair records to the rank that composes the trajectories (the reference's only sequential
// step, src/PUTSLAM/PUTSLAM.cpp:735-740).
//
// Round 6: nothing on the data path joins.  Every member owns a PsBatchQueue (four launch chains) and a communication stream.
// A batch's records are packed by a small kernel queued on each chain right behind that chain's share of the batch (so the
// chain's next batch cannot overwrite a pose before it has been packed), with an event behind it.  The gather itself -- ncclGather,
// the copy to pinned host memory on the root, the ticket's event -- goes onto the communication stream once the HOST has seen that
// event complete: at the next submit / gather call that finds it so, or inside ps_shard_wait, which waits for it (member_flush;
// tickets in order, every rank the same order).  The chains never wait for the communication stream, and the communication stream
// never holds a wait for a chain: a cross-queue wait that stays pending while the chains run costs THEM 5 % at 499 pairs per batch
// and 19 % at 125 (profiles/r06v/pending_waits.txt) -- round 6's first form queued the gather at once behind such a wait and lay
// 4 - 5 % below the batch queue's rate; this form lies 1 - 2 % below it.  A record block is one of PS_SHARD_GATHERS_IN_FLIGHT, and
// it is the HOST that waits (in submit) when all of them are outstanding.  Rounds 1 - 5 packed, gathered, copied and synchronised
// every member on the member's one chain, every call, from one host thread.
//
// One process driving several GPUs: from two members on every member has a host thread of its own; a call that addresses all
// members hands each thread its member's share and waits for all of them.  RCCL collectives are then issued one per thread on
// that thread's communicator (no ncclGroupStart needed); with one member the calls run inline.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "putslam_shard.h"

static_assert(PS_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
static_assert(PS_SHARD_RECORD_FLOATS == PS_RECORD_FLOATS, "record size");

namespace {

constexpr int kSlots = PS_SHARD_GATHERS_IN_FLIGHT;
constexpr int kMaxChains = PS_BATCH_QUEUE_MAX_CHAINS;
constexpr int kProducers = kMaxChains + 1; // the queue's chains + the member's own context stream (the blocking form)

struct Slot { // the buffers of one gather on one member
    float *rec = nullptr;      // device: this member's packed records
    size_t recCap = 0;         // floats
    float *gathered = nullptr; // device, root only: [world][pairs][18]
    size_t gatheredCap = 0;
    float *host = nullptr;     // pinned, root only
    size_t hostCap = 0;
    hipEvent_t done = nullptr; // behind the gather and the copy to the host
    bool pending = false;      // a gather has been queued on this slot and not waited for by the host
    long long ticket = -1;
    int pairsPerRank = 0;
    bool isRoot = false;
    // the gather's request, kept until the records are packed (member_flush): who packs (an event behind every producer's packing
    // kernel), how many pairs this member packed, where the records go
    hipEvent_t packed[PS_BATCH_QUEUE_MAX_CHAINS + 1] = {};
    bool packedUsed[PS_BATCH_QUEUE_MAX_CHAINS + 1] = {};
    bool requested = false;    // ps_shard_gather_records_async has asked for this slot's gather; it has not been queued yet
    int lastP = 0, root = 0;
};

struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> task;
    bool has = false, quit = false;
};

struct Member {
    int device = 0, rank = 0;
    PsContext *ctx = nullptr;
    PsBatchQueue *queue = nullptr; // created at the first submit (the options of ctx at that moment are its chains')
    ncclComm_t comm = nullptr;
    hipStream_t commStream = nullptr;
    uint8_t *blob = nullptr; // device: parameter block
    Slot slot[kSlots];
    long long nextFlush = 0; // the oldest gather that has been asked for and not queued yet (tickets are queued in order)
    std::string err;
    Worker *worker = nullptr;
};

} // namespace

struct PsShardGroup {
    int world = 0;
    std::vector<Member> m;
    std::string err;
    long long nextGather = 0;  // the next ticket
    long long nextRequest = 0; // tickets below this have been asked for (member_flush queues them)
    bool threaded = false;
    // completion of a fan-out over the workers
    std::mutex jm;
    std::condition_variable jcv;
    int remaining = 0;
    std::vector<int> rcs;
};

namespace {

int sfail(PsShardGroup *g, int code, const char *what, const char *detail = nullptr)
{
    if (g) {
        g->err = what;
        if (detail) {
            g->err += ": ";
            g->err += detail;
        }
    }
    return code;
}

int mfail(Member &mb, int code, const char *what, const char *detail = nullptr)
{
    mb.err = what;
    if (detail) {
        mb.err += ": ";
        mb.err += detail;
    }
    return code;
}

#define SH_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) return sfail(g, PS_ERR_HIP, #call, hipGetErrorString(e_)); \
    } while (0)
#define MB_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) return mfail(mb, PS_ERR_HIP, #call, hipGetErrorString(e_)); \
    } while (0)
#define MB_NCCL(call)                                                                    \
    do {                                                                                 \
        ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess) return mfail(mb, PS_ERR_HIP, #call, ncclGetErrorString(r_)); \
    } while (0)

void worker_main(Member *mb)
{
    (void)hipSetDevice(mb->device);
    Worker &w = *mb->worker;
    for (;;) {
        std::function<int()> task;
        {
            std::unique_lock<std::mutex> lk(w.m);
            w.cv.wait(lk, [&] { return w.has || w.quit; });
            if (w.quit && !w.has) return;
            task = std::move(w.task);
            w.has = false;
        }
        task(); // (reports its result and its completion itself: for_members)
    }
}

// fn(local) for every local member: on the members' own threads, side by side, when the group is threaded; inline otherwise.
// Returns the first failing member's status; its text becomes the group's.
int for_members(PsShardGroup *g, const std::function<int(int)> &fn)
{
    const int L = (int)g->m.size();
    g->rcs.assign((size_t)L, PS_OK);
    if (!g->threaded) {
        for (int i = 0; i < L; ++i) {
            (void)hipSetDevice(g->m[(size_t)i].device);
            g->rcs[(size_t)i] = fn(i);
        }
    } else {
        {
            std::lock_guard<std::mutex> lk(g->jm);
            g->remaining = L;
        }
        for (int i = 0; i < L; ++i) {
            Worker &w = *g->m[(size_t)i].worker;
            {
                std::lock_guard<std::mutex> lk(w.m);
                w.task = [g, i, &fn]() {
                    const int rc = fn(i);
                    std::lock_guard<std::mutex> lk2(g->jm);
                    g->rcs[(size_t)i] = rc;
                    if (--g->remaining == 0) g->jcv.notify_all();
                    return rc;
                };
                w.has = true;
            }
            w.cv.notify_one();
        }
        std::unique_lock<std::mutex> lk(g->jm);
        g->jcv.wait(lk, [&] { return g->remaining == 0; });
    }
    for (int i = 0; i < L; ++i)
        if (g->rcs[(size_t)i] != PS_OK) {
            g->err = "rank " + std::to_string(g->m[(size_t)i].rank) + ": " + g->m[(size_t)i].err;
            return g->rcs[(size_t)i];
        }
    return PS_OK;
}

int member_init(PsShardGroup *g, Member &mb)
{
    SH_HIP(hipSetDevice(mb.device));
    int rc = ps_context_create(mb.device, &mb.ctx);
    if (rc != PS_OK) return sfail(g, rc, "ps_context_create");
    SH_HIP(hipMalloc((void **)&mb.blob, sizeof(PsShardRunParams)));
    SH_HIP(hipStreamCreateWithFlags(&mb.commStream, hipStreamNonBlocking));
    for (Slot &s : mb.slot) {
        SH_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        for (hipEvent_t &e : s.packed) SH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return PS_OK;
}

void start_workers(PsShardGroup *g)
{
    // Two members or more: always threaded (one RCCL call per thread on that thread's communicator; a single thread issuing the
    // members' collectives one after the other would need ncclGroupStart / End around them).  PUTSLAM_SHARD_THREADS=1 is a test
    // hook: a worker thread for a single member too, so that the one-GPU boxes exercise the threaded path.
    const char *force = std::getenv("PUTSLAM_SHARD_THREADS");
    g->threaded = g->m.size() >= 2 || (force && std::atoi(force) != 0);
    if (!g->threaded) return;
    for (Member &mb : g->m) {
        mb.worker = new Worker();
        mb.worker->th = std::thread(worker_main, &mb);
    }
}

int find_root(const PsShardGroup *g, int root)
{
    for (size_t i = 0; i < g->m.size(); ++i)
        if (g->m[i].rank == root) return (int)i;
    return -1;
}

int ensure_queue(Member &mb)
{
    if (mb.queue) return PS_OK;
    int rc = ps_batch_queue_create(mb.ctx, 0, &mb.queue); // (the library's default: four chains)
    if (rc != PS_OK) return mfail(mb, rc, "ps_batch_queue_create", ps_last_error(mb.ctx));
    return PS_OK;
}

// the host waits for the slot's previous gather before the slot's blocks are written again
int slot_free(Member &mb, Slot &s)
{
    if (s.pending) {
        MB_HIP(hipEventSynchronize(s.done));
        s.pending = false;
    }
    return PS_OK;
}

// room for `floats` packed records in the slot; what is already packed there is kept (drains the member's producers first)
int ensure_rec(Member &mb, Slot &s, size_t floats, bool keep)
{
    if (floats <= s.recCap) return PS_OK;
    size_t want = floats * 2;
    if (want < (size_t)1024 * PS_SHARD_RECORD_FLOATS) want = (size_t)1024 * PS_SHARD_RECORD_FLOATS;
    float *nw = nullptr;
    MB_HIP(hipMalloc((void **)&nw, want * sizeof(float)));
    if (s.rec) {
        // the old block may be the target of packing launches still queued: drain them
        if (mb.queue && ps_batch_queue_synchronize(mb.queue) != PS_OK) return mfail(mb, PS_ERR_HIP, "ps_batch_queue_synchronize");
        if (ps_context_synchronize(mb.ctx) != PS_OK) return mfail(mb, PS_ERR_HIP, "ps_context_synchronize");
        MB_HIP(hipStreamSynchronize(mb.commStream));
        if (keep && s.recCap) MB_HIP(hipMemcpy(nw, s.rec, s.recCap * sizeof(float), hipMemcpyDeviceToDevice));
        (void)hipFree(s.rec);
    }
    s.rec = nw;
    s.recCap = want;
    return PS_OK;
}

// The gather of slot `s` goes onto the communication stream: its records are packed (the caller has seen every producer's event
// complete), so nothing is queued that has to wait for another queue.
int queue_gather(PsShardGroup *g, Member &mb, Slot &s)
{
    const int pairsPerRank = s.pairsPerRank, root = s.root;
    const size_t recFloats = (size_t)pairsPerRank * PS_SHARD_RECORD_FLOATS;
    int rc = ensure_rec(mb, s, recFloats, true);
    if (rc != PS_OK) return rc;
    if (s.lastP < pairsPerRank)
        MB_HIP(hipMemsetAsync(s.rec + (size_t)s.lastP * PS_SHARD_RECORD_FLOATS, 0, (size_t)(pairsPerRank - s.lastP) * PS_SHARD_RECORD_FLOATS * sizeof(float),
                              mb.commStream));
    s.isRoot = mb.rank == root;
    if (s.isRoot) {
        const size_t all = recFloats * (size_t)g->world;
        if (s.gatheredCap < all) {
            MB_HIP(hipStreamSynchronize(mb.commStream));
            if (s.gathered) (void)hipFree(s.gathered);
            s.gathered = nullptr;
            MB_HIP(hipMalloc((void **)&s.gathered, all * sizeof(float)));
            s.gatheredCap = all;
        }
        if (s.hostCap < all) {
            if (s.host) (void)hipHostFree(s.host);
            s.host = nullptr;
            MB_HIP(hipHostMalloc((void **)&s.host, all * sizeof(float), hipHostMallocDefault));
            s.hostCap = all;
        }
    }
    MB_NCCL(ncclGather(s.rec, s.isRoot ? s.gathered : nullptr, recFloats, ncclFloat, root, mb.comm, mb.commStream));
    if (s.isRoot)
        MB_HIP(hipMemcpyAsync(s.host, s.gathered, recFloats * (size_t)g->world * sizeof(float), hipMemcpyDeviceToHost, mb.commStream));
    MB_HIP(hipEventRecord(s.done, mb.commStream));
    s.pending = true;
    s.requested = false;
    for (bool &u : s.packedUsed) u = false; // (consumed: a later gather on this slot with no submit before it sends zeros)
    s.lastP = 0;
    return PS_OK;
}

// Gathers that have been asked for are queued IN TICKET ORDER (every rank issues its collectives in the same order), each one when
// its records are packed: up to `upto`, waiting for the packing (force) or stopping at the first one that is not packed yet.
//
// Why the host mediates: a gather queued at once had to make the communication stream wait, on the device, for the packing event
// of a batch that runs for milliseconds -- and a cross-queue wait that stays PENDING slows the launch chains themselves:
// demos/cpp/demo_batch_queue --pending-waits 1 (one such wait per batch on a stream of its own) reads 545 k instead of 574 k
// frame-pairs/s at 499 pairs per batch and 395 k instead of 484 k at 125 (profiles/r06v/pending_waits.txt) -- the 4 - 5 % the
// native path lay below the batch queue's rate.  A wait for an event the host has SEEN complete is none.
int member_flush(PsShardGroup *g, Member &mb, long long upto, bool force)
{
    for (;;) {
        const long long t = mb.nextFlush;
        if (t > upto || t >= g->nextRequest) return PS_OK;
        Slot &s = mb.slot[t % kSlots];
        if (!s.requested || s.ticket != t) { // (nothing to queue for this ticket on this member)
            mb.nextFlush = t + 1;
            continue;
        }
        for (int i = 0; i < kProducers; ++i) {
            if (!s.packedUsed[i]) continue;
            if (force) {
                MB_HIP(hipEventSynchronize(s.packed[i]));
            } else {
                const hipError_t e = hipEventQuery(s.packed[i]);
                if (e == hipErrorNotReady) {
                    (void)hipGetLastError();
                    return PS_OK; // not packed yet: this gather and the later ones stay where they are
                }
                if (e != hipSuccess) return mfail(mb, PS_ERR_HIP, "hipEventQuery", hipGetErrorString(e));
            }
        }
        int rc = queue_gather(g, mb, s);
        if (rc != PS_OK) return rc;
        mb.nextFlush = t + 1;
    }
}

// the slot's blocks are about to be written again: whatever the slot still owes is settled first
int slot_retire(PsShardGroup *g, Member &mb, Slot &s)
{
    if (s.requested) { // (the host never waited for that ticket: its gather is queued now, after the older ones)
        int rc = member_flush(g, mb, s.ticket, true);
        if (rc != PS_OK) return rc;
    }
    return slot_free(mb, s);
}

int member_submit(PsShardGroup *g, Member &mb, const PsShardJob &job)
{
    int rc = member_flush(g, mb, g->nextRequest, false); // (whatever has been packed meanwhile)
    if (rc != PS_OK) return rc;
    Slot &s = mb.slot[g->nextGather % kSlots];
    rc = slot_retire(g, mb, s);
    if (rc != PS_OK) return rc;
    for (bool &u : s.packedUsed) u = false;
    s.lastP = 0;
    if (job.P <= 0) return PS_OK;
    if (!job.params || !job.cfg || !job.frames || !job.pairs || !job.out || !job.out->pose || !job.out->stats)
        return mfail(mb, PS_ERR_BAD_ARG, "ps_shard_submit_all: bad job");
    rc = ensure_queue(mb);
    if (rc != PS_OK) return rc;
    rc = ensure_rec(mb, s, (size_t)job.P * PS_SHARD_RECORD_FLOATS, false);
    if (rc != PS_OK) return rc;
    rc = ps_batch_queue_submit(mb.queue, job.params, job.cfg, job.K, job.frames, job.pairs, job.P, job.out, nullptr);
    if (rc != PS_OK) return mfail(mb, rc, "ps_batch_queue_submit", ps_last_error(mb.ctx));
    int32_t bounds[kMaxChains + 1];
    const int C = ps_batch_queue_last_split(mb.queue, bounds);
    for (int i = 0; i < C; ++i) {
        const int lo = bounds[i], n = bounds[i + 1] - bounds[i];
        if (n <= 0) continue;
        PsContext *cc = ps_batch_queue_context(mb.queue, i);
        hipStream_t st = (hipStream_t)ps_context_stream(cc);
        // behind this chain's share of the batch, before the chain's next batch can overwrite a pose
        if (ps_pack_records_device(cc, nullptr, job.out->pose + (size_t)lo * 16, job.out->stats + lo, n, n, s.rec + (size_t)lo * PS_SHARD_RECORD_FLOATS) != PS_OK)
            return mfail(mb, PS_ERR_HIP, "ps_pack_records_device", ps_last_error(cc));
        MB_HIP(hipEventRecord(s.packed[i], st));
        s.packedUsed[i] = true;
    }
    s.lastP = job.P;
    return PS_OK;
}

// the blocking form's producer: results a host wrote with ps_vo_pairs_device on the member's own context
int member_pack_from_context(PsShardGroup *g, Member &mb, const PsPairResults &res, int valid, int pairsPerRank)
{
    if (valid < 0 || valid > pairsPerRank || !res.pose || !res.stats)
        return mfail(mb, PS_ERR_BAD_ARG, "ps_shard_gather_records: bad results / validPairs");
    Slot &s = mb.slot[g->nextGather % kSlots];
    int rc = slot_retire(g, mb, s);
    if (rc != PS_OK) return rc;
    for (bool &u : s.packedUsed) u = false;
    s.lastP = 0;
    rc = ensure_rec(mb, s, (size_t)pairsPerRank * PS_SHARD_RECORD_FLOATS, false);
    if (rc != PS_OK) return rc;
    hipStream_t st = (hipStream_t)ps_context_stream(mb.ctx);
    if (ps_pack_records_device(mb.ctx, nullptr, res.pose, res.stats, valid, pairsPerRank, s.rec) != PS_OK)
        return mfail(mb, PS_ERR_HIP, "ps_pack_records_device", ps_last_error(mb.ctx));
    MB_HIP(hipEventRecord(s.packed[kMaxChains], st));
    s.packedUsed[kMaxChains] = true;
    s.lastP = pairsPerRank; // (zero-filled by the kernel itself)
    return PS_OK;
}

// ps_shard_gather_records_async on one member: the request is noted on the ticket's slot; the gather itself is queued by
// member_flush once the records are packed
int member_request(PsShardGroup *g, Member &mb, long long ticket, int pairsPerRank, int root)
{
    Slot &s = mb.slot[ticket % kSlots];
    int rc = slot_retire(g, mb, s); // (a gather with no submit since this slot's last use; after a submit there is nothing left to settle)
    if (rc != PS_OK) return rc;
    if (s.lastP > pairsPerRank) return mfail(mb, PS_ERR_BAD_ARG, "ps_shard_gather_records_async: pairsPerRank is smaller than the member's batch");
    s.requested = true;
    s.ticket = ticket;
    s.pairsPerRank = pairsPerRank;
    s.root = root;
    return PS_OK;
}

} // namespace

extern "C" {

int ps_shard_group_create(const int *devices, int numDevices, PsShardGroup **out)
{
    if (!out) return PS_ERR_BAD_ARG;
    *out = nullptr;
    int avail = 0;
    if (hipGetDeviceCount(&avail) != hipSuccess || avail <= 0) return PS_ERR_NO_DEVICE; // no CPU fallback
    if (numDevices < 1 || numDevices > 64) return PS_ERR_BAD_ARG;
    std::vector<int> dev((size_t)numDevices);
    for (int i = 0; i < numDevices; ++i) {
        dev[(size_t)i] = devices ? devices[i] : i;
        if (dev[(size_t)i] < 0 || dev[(size_t)i] >= avail) return PS_ERR_BAD_ARG;
    }
    PsShardGroup *g = new PsShardGroup();
    g->world = numDevices;
    g->m.resize((size_t)numDevices);
    std::vector<ncclComm_t> comms((size_t)numDevices, nullptr);
    ncclResult_t r = ncclCommInitAll(comms.data(), numDevices, dev.data());
    if (r != ncclSuccess) {
        fprintf(stderr, "ps_shard_group_create: ncclCommInitAll: %s\n", ncclGetErrorString(r));
        delete g;
        return PS_ERR_HIP;
    }
    for (int i = 0; i < numDevices; ++i) {
        Member &mb = g->m[(size_t)i];
        mb.device = dev[(size_t)i];
        mb.rank = i;
        mb.comm = comms[(size_t)i];
    }
    for (Member &mb : g->m) {
        int rc = member_init(g, mb);
        if (rc != PS_OK) {
            fprintf(stderr, "ps_shard_group_create: %s\n", g->err.c_str());
            ps_shard_group_destroy(g);
            return rc;
        }
    }
    start_workers(g);
    *out = g;
    return PS_OK;
}

int ps_shard_unique_id(uint8_t id[PS_SHARD_ID_BYTES])
{
    if (!id) return PS_ERR_BAD_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return PS_ERR_HIP;
    memcpy(id, u.internal, PS_SHARD_ID_BYTES);
    return PS_OK;
}

int ps_shard_group_create_rank(int device, int rank, int worldSize, const uint8_t id[PS_SHARD_ID_BYTES], PsShardGroup **out)
{
    if (!out || !id || worldSize < 1 || rank < 0 || rank >= worldSize) return PS_ERR_BAD_ARG;
    *out = nullptr;
    int avail = 0;
    if (hipGetDeviceCount(&avail) != hipSuccess || avail <= 0) return PS_ERR_NO_DEVICE;
    if (device < 0 || device >= avail) return PS_ERR_BAD_ARG;
    if (hipSetDevice(device) != hipSuccess) return PS_ERR_HIP;
    PsShardGroup *g = new PsShardGroup();
    g->world = worldSize;
    g->m.resize(1);
    Member &mb = g->m[0];
    mb.device = device;
    mb.rank = rank;
    ncclUniqueId u;
    memcpy(u.internal, id, PS_SHARD_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&mb.comm, worldSize, u, rank);
    if (r != ncclSuccess) {
        fprintf(stderr, "ps_shard_group_create_rank: ncclCommInitRank: %s\n", ncclGetErrorString(r));
        mb.comm = nullptr;
        delete g;
        return PS_ERR_HIP;
    }
    int rc = member_init(g, mb);
    if (rc != PS_OK) {
        fprintf(stderr, "ps_shard_group_create_rank: %s\n", g->err.c_str());
        ps_shard_group_destroy(g);
        return rc;
    }
    start_workers(g);
    *out = g;
    return PS_OK;
}

void ps_shard_group_destroy(PsShardGroup *g)
{
    if (!g) return;
    for (Member &mb : g->m)
        if (mb.worker) {
            {
                std::lock_guard<std::mutex> lk(mb.worker->m);
                mb.worker->quit = true;
            }
            mb.worker->cv.notify_one();
            if (mb.worker->th.joinable()) mb.worker->th.join();
            delete mb.worker;
            mb.worker = nullptr;
        }
    for (Member &mb : g->m) {
        (void)hipSetDevice(mb.device);
        if (mb.queue) (void)ps_batch_queue_synchronize(mb.queue);
        if (mb.ctx) (void)ps_context_synchronize(mb.ctx);
        if (mb.commStream) (void)hipStreamSynchronize(mb.commStream);
        if (mb.comm) (void)ncclCommDestroy(mb.comm);
        for (Slot &s : mb.slot) {
            if (s.rec) (void)hipFree(s.rec);
            if (s.gathered) (void)hipFree(s.gathered);
            if (s.host) (void)hipHostFree(s.host);
            if (s.done) (void)hipEventDestroy(s.done);
            for (hipEvent_t e : s.packed)
                if (e) (void)hipEventDestroy(e);
        }
        if (mb.blob) (void)hipFree(mb.blob);
        if (mb.commStream) (void)hipStreamDestroy(mb.commStream);
        if (mb.queue) ps_batch_queue_destroy(mb.queue);
        if (mb.ctx) ps_context_destroy(mb.ctx);
    }
    delete g;
}

const char *ps_shard_last_error(const PsShardGroup *g) { return g ? g->err.c_str() : "null group"; }
int ps_shard_world_size(const PsShardGroup *g) { return g ? g->world : (int)PS_ERR_BAD_ARG; }
int ps_shard_local_count(const PsShardGroup *g) { return g ? (int)g->m.size() : (int)PS_ERR_BAD_ARG; }
int ps_shard_rank(const PsShardGroup *g, int local)
{
    return (g && local >= 0 && local < (int)g->m.size()) ? g->m[(size_t)local].rank : (int)PS_ERR_BAD_ARG;
}
int ps_shard_device(const PsShardGroup *g, int local)
{
    return (g && local >= 0 && local < (int)g->m.size()) ? g->m[(size_t)local].device : (int)PS_ERR_BAD_ARG;
}
PsContext *ps_shard_context(PsShardGroup *g, int local)
{
    return (g && local >= 0 && local < (int)g->m.size()) ? g->m[(size_t)local].ctx : nullptr;
}
PsBatchQueue *ps_shard_queue(PsShardGroup *g, int local)
{
    if (!g || local < 0 || local >= (int)g->m.size()) return nullptr;
    Member &mb = g->m[(size_t)local];
    (void)hipSetDevice(mb.device);
    return ensure_queue(mb) == PS_OK ? mb.queue : nullptr;
}

void ps_shard_range(int64_t total, int world, int rank, int64_t *lo, int64_t *hi)
{
    if (world < 1) world = 1;
    const int64_t base = total / world, rem = total % world;
    const int64_t l = rank * base + (rank < rem ? rank : rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

int ps_shard_broadcast_params(PsShardGroup *g, PsShardRunParams *perLocal, int root)
{
    if (!g || !perLocal || root < 0 || root >= g->world) return sfail(g, PS_ERR_BAD_ARG, "ps_shard_broadcast_params: bad argument");
    return for_members(g, [g, perLocal, root](int i) -> int {
        Member &mb = g->m[(size_t)i];
        if (mb.rank == root) MB_HIP(hipMemcpyAsync(mb.blob, &perLocal[i], sizeof(PsShardRunParams), hipMemcpyHostToDevice, mb.commStream));
        MB_NCCL(ncclBroadcast(mb.blob, mb.blob, sizeof(PsShardRunParams), ncclUint8, root, mb.comm, mb.commStream));
        MB_HIP(hipMemcpyAsync(&perLocal[i], mb.blob, sizeof(PsShardRunParams), hipMemcpyDeviceToHost, mb.commStream));
        MB_HIP(hipStreamSynchronize(mb.commStream));
        return PS_OK;
    });
}

int ps_shard_submit_all(PsShardGroup *g, const PsShardJob *jobs)
{
    if (!g || !jobs) return sfail(g, PS_ERR_BAD_ARG, "ps_shard_submit_all: bad argument");
    return for_members(g, [g, jobs](int i) -> int { return member_submit(g, g->m[(size_t)i], jobs[i]); });
}

int ps_shard_gather_records_async(PsShardGroup *g, int pairsPerRank, int root, int64_t *ticket)
{
    if (ticket) *ticket = -1;
    if (!g || pairsPerRank < 1 || root < 0 || root >= g->world) return sfail(g, PS_ERR_BAD_ARG, "ps_shard_gather_records_async: bad argument");
    const long long t = g->nextGather;
    g->nextRequest = t + 1; // (read by member_flush on the members' threads: set before they start)
    // (whatever is packed already goes out now; the rest at the next submit, gather or wait)
    int rc = for_members(g, [g, t, pairsPerRank, root](int i) -> int {
        int rc2 = member_request(g, g->m[(size_t)i], t, pairsPerRank, root);
        return rc2 != PS_OK ? rc2 : member_flush(g, g->m[(size_t)i], t, false);
    });
    if (rc != PS_OK) {
        g->nextRequest = t; // (the ticket was not given out; a member that did note the request forgets it at its slot's next use)
        for (Member &mb : g->m) {
            Slot &s = mb.slot[t % kSlots];
            if (s.requested && s.ticket == t) s.requested = false;
        }
        return rc;
    }
    g->nextGather = t + 1;
    if (ticket) *ticket = t;
    return PS_OK;
}

int ps_shard_wait(PsShardGroup *g, int64_t ticket, const float **hostRecords)
{
    if (hostRecords) *hostRecords = nullptr;
    if (!g || ticket < 0 || ticket >= g->nextGather) return sfail(g, PS_ERR_BAD_ARG, "ps_shard_wait: no such ticket");
    // the gathers up to this ticket that are not queued yet are queued now (the host waits for their records to be packed)
    const long long upto = ticket;
    int rc = for_members(g, [g, upto](int i) -> int { return member_flush(g, g->m[(size_t)i], upto, true); });
    if (rc != PS_OK) return rc;
    for (Member &mb : g->m) {
        Slot &s = mb.slot[ticket % kSlots];
        if (s.ticket != ticket) continue; // the slot has moved on: that gather is complete, its records are gone
        if (s.pending) {
            SH_HIP(hipSetDevice(mb.device));
            SH_HIP(hipEventSynchronize(s.done));
            s.pending = false;
        }
        if (s.isRoot && hostRecords) *hostRecords = s.host;
    }
    return PS_OK;
}

int ps_shard_gather_records(PsShardGroup *g, const PsPairResults *results, const int32_t *validPairs, int pairsPerRank,
                            float *hostRecords, int root)
{
    if (!g || !results || pairsPerRank < 0 || root < 0 || root >= g->world)
        return sfail(g, PS_ERR_BAD_ARG, "ps_shard_gather_records: bad argument");
    if (pairsPerRank == 0) return PS_OK;
    const int rl = find_root(g, root);
    if (rl >= 0 && !hostRecords) return sfail(g, PS_ERR_BAD_ARG, "ps_shard_gather_records: this process drives the root and needs hostRecords");
    int rc = for_members(g, [g, results, validPairs, pairsPerRank](int i) -> int {
        return member_pack_from_context(g, g->m[(size_t)i], results[i], validPairs ? validPairs[i] : pairsPerRank, pairsPerRank);
    });
    if (rc != PS_OK) return rc;
    int64_t t = -1;
    rc = ps_shard_gather_records_async(g, pairsPerRank, root, &t);
    if (rc != PS_OK) return rc;
    const float *rec = nullptr;
    rc = ps_shard_wait(g, t, &rec);
    if (rc != PS_OK) return rc;
    if (rl >= 0 && rec) memcpy(hostRecords, rec, (size_t)g->world * pairsPerRank * PS_SHARD_RECORD_FLOATS * sizeof(float));
    return PS_OK;
}

int ps_shard_synchronize(PsShardGroup *g)
{
    if (!g) return PS_ERR_BAD_ARG;
    if (g->nextRequest > 0) { // (every gather that has been asked for is queued first)
        const long long upto = g->nextRequest - 1;
        int rc = for_members(g, [g, upto](int i) -> int { return member_flush(g, g->m[(size_t)i], upto, true); });
        if (rc != PS_OK) return rc;
    }
    for (Member &mb : g->m) {
        SH_HIP(hipSetDevice(mb.device));
        if (mb.queue && ps_batch_queue_synchronize(mb.queue) != PS_OK) return sfail(g, PS_ERR_HIP, "ps_batch_queue_synchronize", ps_last_error(mb.ctx));
        if (ps_context_synchronize(mb.ctx) != PS_OK) return sfail(g, PS_ERR_HIP, "ps_context_synchronize", ps_last_error(mb.ctx));
        SH_HIP(hipStreamSynchronize(mb.commStream));
    }
    return PS_OK;
}

} // extern "C"
