// ps_batch_queue.cpp -- PsBatchQueue (include/putslam_hip.h): the submission pattern that gives a looping host the library's
// full rate, inside the library.
//
// A host that hands batch after batch to ONE context gets one launch chain: the matrix-core Hamming sweep and the vector scoring
// stages of a batch run one after the other, dependent launches with the chip partly idle between them (517 k frame-pairs/s on
// the 499-pair workload from a C++ loop).  Launch chains that are never joined overlap one batch's Hamming sweep with another's
// scoring.  Rounds 3 - 5 got that by SPLITTING every batch 45 % / 55 % over two contexts + streams (bench.py's submission: 559 k);
// measured in round 6 from demos/cpp/demo_batch_queue on the same data, handing WHOLE batches to the chains in turn is better at
// every batch size tried (profiles/r06h/queue_split_vs_turns.txt):
//       pairs per batch        64      125      250      499     1000        499 (bench sequence)   499, E0 / RANSAC <= 487
//       one context         142 k    274 k    336 k    451 k    496 k        517 k
//       split 45 / 55       160 k    175 k    412 k    482 k    547 k        559 k                   1.85 M
//       whole, in turn      180 k    402 k    473 k    543 k    562 k        608 k                   2.17 M
// -- a batch keeps its full launch shapes (half the launches per pair, grids twice as large; and a sub-batch of a small batch
// drops below the staged scoring's cost-model threshold: 125 pairs split run complete scoring, twice the work) while the chains
// still drift out of step.  So that is what the queue does: batch n runs on chain n mod chains, whole.  The split form is kept
// behind PUTSLAM_HIP_QUEUE_SPLIT_FROM=<pairs> (batches of at least that many pairs are split, two chains: 45 % / 55 %) for A/B runs.
// (The table is two chains'.  FOUR are the best count at every batch size from 16 to 1000 pairs and the default since late round 6 --
// 499 pairs: 610 k on the bench sequence, 125: 490 k, 64: 416 k with the chains' side_by_side option, which moves the staged
// scoring's crossover to where it lies when other chains fill the gaps between its launches: profiles/r06u, profiles/r06v.)
//
//   * chains are ordered only within themselves: submit never makes one chain wait for another, and a batch's completion is an
//     event per chain it ran on, recorded behind its last launch -- waited for by the host (ps_batch_queue_wait) or by a stream of
//     the host's (ps_batch_queue_wait_on_stream), never by another chain;
//   * pair p of a batch keeps its hypothesis stream cfg->seed + p: results are byte for byte those of one ps_vo_pairs_device call;
//   * consecutive batches run side by side: they need output blocks of their own (two blocks used in turn do for two chains).
// Host-only translation unit: everything here is queue bookkeeping around ps_vo_pairs_device.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ps_internal.h"

namespace {

constexpr int kMaxChains = PS_BATCH_QUEUE_MAX_CHAINS;
constexpr int kDefaultChains = 4; // (profiles/r06u/small_batch_chains.txt: the best count at every batch size tried)
constexpr int kTickets = 64;      // batches that can be in flight; submit waits for the oldest when the ring is full
constexpr int kSplitFromPairs = 0x7fffffff; // (never: whole batches, the chains in turn; PUTSLAM_HIP_QUEUE_SPLIT_FROM overrides)

struct Ticket {
    long long seq = -1;              // batch number this slot holds (-1: never used)
    bool used[kMaxChains] = {};
    hipEvent_t ev[kMaxChains] = {};
};

} // namespace

struct PsBatchQueue {
    PsContext *parent = nullptr;     // options' source and the place error texts go to (not owned)
    int device = 0;
    int chains = 0;
    int splitPermille = 450;         // two chains: share of the pairs on chain 0
    int splitFrom = kSplitFromPairs; // batches of fewer pairs go to the chains in turn, whole
    PsContext *ctx[kMaxChains] = {};
    Ticket ring[kTickets];
    long long next = 0;              // number of the next batch
    int32_t lastBounds[kMaxChains + 1] = {};
    // the output block (its pose array) and the number of the last whole batch every chain was given: a batch that writes the block
    // of a batch still in flight on ANOTHER chain follows it on that chain instead of racing it
    const void *lastOut[kMaxChains] = {};
    long long lastSeq[kMaxChains];
    PsBatchQueue()
    {
        for (long long &x : lastSeq) x = -1;
    }
};

namespace {

int qfail(PsBatchQueue *q, int code, const std::string &what)
{
    if (q && q->parent) psi_set_error(q->parent, what.c_str());
    return code;
}

int wait_slot(PsBatchQueue *q, Ticket &t)
{
    for (int i = 0; i < q->chains; ++i)
        if (t.used[i]) {
            hipError_t e = hipEventSynchronize(t.ev[i]);
            if (e != hipSuccess) return qfail(q, PS_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
        }
    return PS_OK;
}

// which pairs of a batch of P go to which chain: bounds[0 .. chains]
void split(const PsBatchQueue *q, int P, long long seq, int32_t *bounds)
{
    const int C = q->chains;
    if (C == 1) {
        bounds[0] = 0;
        bounds[1] = P;
        return;
    }
    if (P < q->splitFrom) { // the whole batch on one chain, the chains in turn
        const int c = (int)(seq % C);
        for (int i = 0; i <= C; ++i) bounds[i] = i <= c ? 0 : P;
        return;
    }
    if (C == 2) {
        bounds[0] = 0;
        bounds[1] = (int32_t)((long long)P * q->splitPermille / 1000);
        bounds[2] = P;
        return;
    }
    for (int i = 0; i <= C; ++i) bounds[i] = (int32_t)((long long)P * i / C);
}

} // namespace

extern "C" {

int ps_batch_queue_create(PsContext *ctx, int chains, PsBatchQueue **out)
{
    if (!out) return PS_ERR_BAD_ARG;
    *out = nullptr;
    if (!ctx) return PS_ERR_BAD_ARG;
    if (chains == 0) chains = kDefaultChains;
    if (chains < 1 || chains > kMaxChains) {
        psi_set_error(ctx, "ps_batch_queue_create: chains 1 .. 8 (0 = 4)");
        return PS_ERR_BAD_ARG;
    }
    PsBatchQueue *q = new PsBatchQueue();
    q->parent = ctx;
    q->device = ps_context_device(ctx);
    q->chains = chains;
    if (const char *v = std::getenv("PUTSLAM_HIP_QUEUE_SPLIT")) { // (A/B hook: share of chain 0 in permille)
        const int x = std::atoi(v);
        if (x >= 50 && x <= 950) q->splitPermille = x;
    }
    if (const char *v = std::getenv("PUTSLAM_HIP_QUEUE_SPLIT_FROM")) { // (A/B hook)
        const int x = std::atoi(v);
        if (x >= 2) q->splitFrom = x;
    }
    for (int i = 0; i < chains; ++i) {
        int rc = ps_context_create(q->device, &q->ctx[i]);
        if (rc != PS_OK) {
            q->ctx[i] = nullptr;
            ps_batch_queue_destroy(q);
            psi_set_error(ctx, "ps_batch_queue_create: chain context");
            return rc;
        }
        psi_copy_options(q->ctx[i], ctx);
        // batches on different chains overlap: the staged scoring pays from far smaller batches on (prepare_score)
        if (chains > 1 && ps_context_get_option(q->ctx[i], "side_by_side") < chains) (void)ps_context_set_option(q->ctx[i], "side_by_side", chains);
    }
    if (hipSetDevice(q->device) != hipSuccess) {
        ps_batch_queue_destroy(q);
        psi_set_error(ctx, "ps_batch_queue_create: hipSetDevice");
        return PS_ERR_HIP;
    }
    for (Ticket &t : q->ring)
        for (int i = 0; i < chains; ++i)
            if (hipEventCreateWithFlags(&t.ev[i], hipEventDisableTiming) != hipSuccess) {
                t.ev[i] = nullptr;
                ps_batch_queue_destroy(q);
                psi_set_error(ctx, "ps_batch_queue_create: hipEventCreateWithFlags");
                return PS_ERR_HIP;
            }
    // every chain wants a hardware queue of its own (ps_env.cpp): with fewer the chains share one and serialise -- results are
    // the same, the rate is that of one chain or worse.  Not an error; the text is there for whoever asks.
    if (chains > 1 && psi_hw_queues_seen() > 0 && psi_hw_queues_seen() < chains + 2) {
        char msg[256];
        std::snprintf(msg, sizeof msg,
                      "ps_batch_queue_create: GPU_MAX_HW_QUEUES=%d in the environment: %d chains want %d hardware queues (16 recommended); "
                      "chains that share a queue run one after the other",
                      psi_hw_queues_seen(), chains, chains + 2);
        psi_set_error(ctx, msg);
    }
    *out = q;
    return PS_OK;
}

void ps_batch_queue_destroy(PsBatchQueue *q)
{
    if (!q) return;
    (void)hipSetDevice(q->device);
    for (int i = 0; i < kMaxChains; ++i)
        if (q->ctx[i]) (void)ps_context_synchronize(q->ctx[i]);
    for (Ticket &t : q->ring)
        for (hipEvent_t &e : t.ev)
            if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < kMaxChains; ++i)
        if (q->ctx[i]) ps_context_destroy(q->ctx[i]);
    delete q;
}

int ps_batch_queue_chains(const PsBatchQueue *q) { return q ? q->chains : (int)PS_ERR_BAD_ARG; }

PsContext *ps_batch_queue_context(PsBatchQueue *q, int chain)
{
    return (q && chain >= 0 && chain < q->chains) ? q->ctx[chain] : nullptr;
}

int ps_batch_queue_submit(PsBatchQueue *q, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                          const PsFrameSet *frames, const int32_t *pairs, int P, const PsPairResults *out, int64_t *ticket)
{
    if (!q) return PS_ERR_BAD_ARG;
    if (ticket) *ticket = -1;
    if (!params || !cfg || !frames || !out || P < 0 || (P > 0 && !pairs))
        return qfail(q, PS_ERR_BAD_ARG, "ps_batch_queue_submit: bad argument");
    if (cfg->sampleIdx) return qfail(q, PS_ERR_BAD_ARG, "ps_batch_queue_submit: explicit sample streams are per call, not per batch");
    if (hipSetDevice(q->device) != hipSuccess) return qfail(q, PS_ERR_HIP, "ps_batch_queue_submit: hipSetDevice");
    const long long seq = q->next;
    Ticket &t = q->ring[seq % kTickets];
    if (t.seq >= 0) { // flow control: the slot's previous batch (kTickets batches ago) has to be complete
        int rc = wait_slot(q, t);
        if (rc != PS_OK) return rc;
    }
    int32_t bounds[kMaxChains + 1];
    split(q, P, seq, bounds);
    if (q->chains > 1 && P > 0 && P < q->splitFrom) {
        // Whole batches run side by side on different chains, so two of them must not write the same output block.  A host that
        // hands over the block of a batch that is still in flight on another chain gets stream order instead of a race: the new
        // batch follows that one on ITS chain (and the pair run one after the other: blocks of their own, in turn, is how a
        // host gets the queue's rate).
        int mine = 0;
        for (int i = 0; i < q->chains; ++i)
            if (bounds[i + 1] > bounds[i]) mine = i;
        for (int j = 0; j < q->chains; ++j) {
            if (j == mine || q->lastOut[j] != (const void *)out->pose || q->lastSeq[j] < 0) continue;
            Ticket &tj = q->ring[q->lastSeq[j] % kTickets];
            if (tj.seq != q->lastSeq[j] || !tj.used[j]) continue; // (older than the ring: complete)
            const hipError_t e = hipEventQuery(tj.ev[j]);
            if (e == hipErrorNotReady) {
                (void)hipGetLastError();
                for (int i = 0; i <= q->chains; ++i) bounds[i] = i <= j ? 0 : P;
                mine = j;
                break;
            }
        }
        q->lastOut[mine] = (const void *)out->pose;
        q->lastSeq[mine] = seq;
    }
    const size_t cap = (size_t)frames->maxKpts;
    bool used[kMaxChains] = {};
    int rcAll = PS_OK;
    for (int i = 0; i < q->chains; ++i) {
        const int lo = bounds[i], hi = bounds[i + 1];
        if (hi <= lo) continue;
        PsRansacConfig c2 = *cfg;
        c2.seed = cfg->seed + (uint64_t)lo; // pair p draws from cfg->seed + p on whatever chain it runs
        PsPairResults o2;
        o2.matches = out->matches ? out->matches + (size_t)lo * cap : nullptr;
        o2.numMatches = out->numMatches ? out->numMatches + lo : nullptr;
        o2.inlierMask = out->inlierMask ? out->inlierMask + (size_t)lo * cap : nullptr;
        o2.pose = out->pose ? out->pose + (size_t)lo * 16 : nullptr;
        o2.stats = out->stats ? out->stats + lo : nullptr;
        int rc = ps_vo_pairs_device(q->ctx[i], params, &c2, K, frames, pairs + 2 * (size_t)lo, hi - lo, &o2);
        if (rc != PS_OK) {
            // what was queued stays queued (the other chain's share of this batch may complete); the batch as a whole failed
            rcAll = qfail(q, rc, std::string("ps_batch_queue_submit: chain ") + std::to_string(i) + ": " + ps_last_error(q->ctx[i]));
            break;
        }
        if (hipEventRecord(t.ev[i], (hipStream_t)ps_context_stream(q->ctx[i])) != hipSuccess) {
            (void)ps_context_synchronize(q->ctx[i]); // no event to wait on: drain the chain instead
            continue;
        }
        used[i] = true;
    }
    // (the ticket exists even for a failed batch: waiting for it waits for whatever part of it was queued)
    t.seq = seq;
    for (int i = 0; i < kMaxChains; ++i) t.used[i] = used[i];
    for (int i = 0; i <= q->chains; ++i) q->lastBounds[i] = bounds[i];
    q->next = seq + 1;
    if (ticket) *ticket = seq;
    return rcAll;
}

int ps_batch_queue_last_split(const PsBatchQueue *q, int32_t *bounds)
{
    if (!q || !bounds) return PS_ERR_BAD_ARG;
    for (int i = 0; i <= q->chains; ++i) bounds[i] = q->lastBounds[i];
    return q->chains;
}

static Ticket *find_ticket(PsBatchQueue *q, int64_t ticket, int *rc)
{
    *rc = PS_OK;
    if (ticket < 0 || ticket >= q->next) {
        *rc = qfail(q, PS_ERR_BAD_ARG, "ps_batch_queue: no such ticket");
        return nullptr;
    }
    Ticket &t = q->ring[ticket % kTickets];
    if (t.seq != ticket) return nullptr; // older than the ring: complete (submit waited for it before reusing the slot)
    return &t;
}

int ps_batch_queue_wait(PsBatchQueue *q, int64_t ticket)
{
    if (!q) return PS_ERR_BAD_ARG;
    int rc;
    Ticket *t = find_ticket(q, ticket, &rc);
    if (!t) return rc;
    return wait_slot(q, *t);
}

int ps_batch_queue_query(PsBatchQueue *q, int64_t ticket)
{
    if (!q) return PS_ERR_BAD_ARG;
    int rc;
    Ticket *t = find_ticket(q, ticket, &rc);
    if (!t) return rc == PS_OK ? 1 : rc;
    for (int i = 0; i < q->chains; ++i)
        if (t->used[i]) {
            hipError_t e = hipEventQuery(t->ev[i]);
            if (e == hipErrorNotReady) {
                (void)hipGetLastError();
                return 0;
            }
            if (e != hipSuccess) return qfail(q, PS_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
        }
    return 1;
}

int ps_batch_queue_wait_on_stream(PsBatchQueue *q, int64_t ticket, void *hipStream)
{
    if (!q) return PS_ERR_BAD_ARG;
    int rc;
    Ticket *t = find_ticket(q, ticket, &rc);
    if (!t) return rc;
    if (hipSetDevice(q->device) != hipSuccess) return qfail(q, PS_ERR_HIP, "ps_batch_queue_wait_on_stream: hipSetDevice");
    for (int i = 0; i < q->chains; ++i)
        if (t->used[i]) {
            hipError_t e = hipStreamWaitEvent((hipStream_t)hipStream, t->ev[i], 0);
            if (e != hipSuccess) return qfail(q, PS_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
        }
    return PS_OK;
}

int ps_batch_queue_synchronize(PsBatchQueue *q)
{
    if (!q) return PS_ERR_BAD_ARG;
    for (int i = 0; i < q->chains; ++i) {
        int rc = ps_context_synchronize(q->ctx[i]);
        if (rc != PS_OK) return qfail(q, rc, std::string("ps_batch_queue_synchronize: ") + ps_last_error(q->ctx[i]));
    }
    return PS_OK;
}

} // extern "C"
