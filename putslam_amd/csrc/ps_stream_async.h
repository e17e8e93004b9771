// ps_stream_async.h -- the PIPELINED streaming form of Matcher::match (include/putslam_hip.h: ps_vo_stream_configure_async,
// _push_async, _push_many, _flush, _pop_many, _pop).  Included at the end of ps_capi.hip (it uses that file's PsContext,
// PsVoStream and PS_HIP / PS_ENSURE).
//
// Call shape served: reference src/Matcher/matcher.cpp:452-516 (one frame per call, the previous frame kept as state) in the
// loop of src/PUTSLAM/PUTSLAM.cpp:677-740.  The synchronous ps_vo_stream_push pays a copy in, four kernels, a copy out and a
// synchronisation per frame (0.08 ms: 12.5 k frames/s); here frames are collected into chunks, each chunk is one batched call
// (ps_vo_pairs_device's launches) on one of several lanes, and the results are returned with a lag:
//
//   host frames --(copy stream: SDMA uploads)--> ring of frames in HBM --(lane k % lanes: kernels 1-4)--(download stream)--> pinned block --> pop
//
//   * the copy stream carries ONLY the two large uploads of every chunk (descriptors, points), which the runtime gives to an
//     SDMA engine: back to back they keep the link at 48 GB/s of its 55.  The chunk's small meta block (pair list, row counts)
//     is fetched by a one-work-group kernel on the lane's own stream from mapped pinned memory: as a third hipMemcpyAsync on the
//     copy stream it was a blit kernel between SDMA transfers, and every change of engine left the link idle for 60 us
//     (rocprofv3 timeline, profiles/r05a/stream_trace).  Uploads as kernels over mapped pinned memory reach the link rate too
//     (profiles/microbench/h2d_kernel.hip) but slow the kernels they run beside by 3 - 10 x (profiles/r05b/stream_trace: their
//     outstanding host reads fill the L2's request queues); SDMA transfers do not.
//
//   * places: a chunk needs a place (meta block, device + pinned result blocks, events), not a lane: there are lanes + `ahead`
//     of them (option "stream_ahead"; default: six places in all), and chunk k is LAUNCHED at once on lane k % lanes -- behind that lane's running
//     chunk if it has one.  With one place per lane a lane idled from the end of its chunk's last kernel through the download,
//     the host's pop, the next chunk's upload and the copy-stream -> lane dependency (300 - 500 us of a chunk's ~ 800,
//     profiles/r05h/timeline_ahead0.txt); now its next chunk's launches are already in its stream.
//   * ring: (lanes + ahead + 2) x chunkFrames frame slots.  A chunk's frames are contiguous in it (a chunk that would not fit
//     before the end starts at slot 0 again); the frame before the chunk's first one is still resident, so pair (previous
//     chunk's last frame, this chunk's first frame) needs no second upload.  The chunks alive (running or queued on a lane) read at most the
//     (lanes + ahead - 1) x chunkFrames + 1 slots written last, a jump to slot 0 skips fewer than chunkFrames, the new chunk
//     writes at most chunkFrames: no slot that is still read is overwritten, with no device-side wait.
//   * lane = a private PsContext (stream + scratch arena).  Consecutive chunks go to consecutive lanes, so one chunk's matrix-core
//     Hamming sweep runs beside another's vector scoring sweep, as bench.py's sub-batch chains do.  The pinned result block of a
//     place changes hands at the pop: the caller reads it until the next pop, the place continues with the spare one -- so a
//     place is free the moment its results are popped.
//   * per chunk: 2 uploads, the meta kernel, the batched call's launches, 1 download (5 for a partly filled chunk); uploads are
//     in stream order, so the halo frame needs no event of its own.
//   * results are those of ONE ps_vo_pairs_device call over the whole sequence: pair k draws from cfg->seed + k.
#pragma once

// One chunk's place in the pipeline: meta block, device and pinned result blocks, events.  There are lanes + ahead of them; a
// chunk takes the next one and runs on the next LANE (a private PsContext: stream + scratch arena) in turn, so each lane's
// stream can hold a chunk that runs and further ones queued behind it.
struct AsyncLane {
    PsContext *ctx = nullptr; // the lane context this place's chunk was launched on (not owned)
    Buf meta;                 // device: int32 [2 * B] pair list, then [ringFrames] row counts
    int32_t *hmeta = nullptr; // pinned mirror of it
    Buf res;                  // device: [matches B x cap x 16][mask B x cap][pose B x 64][stats B x 40][numMatches B x 4]
    uint8_t *hres = nullptr;  // pinned mirror of it
    int32_t *hmetaDev = nullptr; // device views of the pinned blocks (hipHostGetDevicePointer)
    uint8_t *hresDev = nullptr;
    hipEvent_t evRun = nullptr, evDone = nullptr; // behind the chunk's kernels / its download
    // small chunks (PsVoAsync::mini): the place's private frames [1 + B] in the packed layout (slot 0 = the frame before the
    // chunk's first one), its chunk as a captured graph, and the arena generation of its lane the graph's pointers belong to
    Buf frames;
    hipGraphExec_t gexec = nullptr;
    unsigned long long gexecGen = 0;
    Plan plan;                  // the plan of the place's full chunk (parameters are fixed per configure), valid while ...
    unsigned long long planGen = 0; // ... its lane's arena is the one it was made with (0 = none)
    int state = 0;            // 0 free, 1 chunk in flight
    long long firstPair = 0;
    int pairs = 0;
    int epoch = 0;
};

// A chunk whose frames are in the ring (or on their way: `up` is recorded behind its uploads), about to be launched.
struct AsyncChunk {
    int pos0 = 0, n = 0;       // ring slots [pos0, pos0 + n)
    int first = 0;             // 1: the chunk's first frame has no predecessor (the stream's first frame, or after a reset)
    int prevPos = -1;          // slot of the frame before the chunk's first one
    long long firstPair = 0;   // index of its first pair in the stream (-> seed)
    int epoch = 0;
    hipEvent_t up = nullptr;   // (owned by PsVoAsync::upEv)
};

struct PsVoAsync {
    int B = 0, lanes = 0, ringFrames = 0;
    int ahead = 0;                        // places beyond one per lane: chunks queued on the lanes' streams behind the running ones
    std::vector<PsContext *> laneCtx;     // the lanes (owned)
    long long launchSeq = 0;              // chunks launched so far (chunk k runs on lane k % lanes)
    long long diagLaunches = 0;           // launches attempted since configure (fault injection of -DPS_STREAM_DIAG builds counts these)
    std::vector<hipEvent_t> upEv;         // lanes + ahead + 1 events, one per chunk alive, by chunk number
    std::vector<uint8_t *> stagePool;     // lanes + ahead + 1 pinned staging areas for frames that arrive one at a time / in
                                          // pageable memory: [B x cap x 32 descriptors][B x cap x 12 points], by chunk number;
                                          // allocated when first needed
    long long chunkSeq = 0;               // chunks uploaded so far
    PsRansacParams prm{};
    PsRansacConfig cfg{};
    float K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool haveK = false;
    Buf ringDesc, ringPts;
    // Small chunks (chunkFrames <= 4, the reference's own call shape at 1: src/Matcher/matcher.cpp:452-516): a chunk is launch-bound
    // -- two uploads, a meta kernel, six or seven launches, an event pair and a download, 85 us of host and queue time for 55 us of
    // kernels: round 5's pipeline was no faster than the synchronous push there.  In this form every place owns a private frame set
    // and its whole chunk is ONE captured hipGraph (frames in by a copy kernel from the pinned staging area, kernels 1 - 4, results
    // out by a kernel into the place's pinned block); what changes between chunks -- row counts, the seed, where the frames lie --
    // travels as data (psdev::MiniMeta).  No ring, no copy streams, no cross-lane dependency: the halo frame is read again from
    // the previous chunk's staging slot.
    bool mini = false;
    // Replaying a place's chunk from its graph is OFF by default: on this runtime hipGraphLaunch of the seven-node graph costs the
    // host more than the seven launches and the lanes' graphs run less side by side -- 20.9 k frames/s at 322 us of lag against
    // 25.1 k at 209 us with ordinary launches (demos/cpp/demo_latency (d), profiles/r06h/mini_chunks.txt).  PUTSLAM_HIP_STREAM_GRAPH=1
    // turns it on (tests run both).
    bool miniGraphs = false;
    std::vector<uint8_t *> stagePoolDev;  // device views of the staging areas (mini chunks read them from kernels)
    const uint8_t *haloDev = nullptr;     // device view of the stream's latest frame in its staging slot (or in the caller's pinned memory)
    const uint8_t *inPlaceDev = nullptr;  // set for ONE mini_submit_body call: the chunk's frames lie in the caller's pinned memory
    uint8_t *loneHost = nullptr, *loneDev = nullptr; // a lone first frame's own pinned block (it takes no place, so no staging area)
    hipEvent_t loneReadEv = nullptr;      // behind the chunk that read that block as its previous frame
    bool loneReadPending = false, haloIsLone = false;
    int haloNk = 0;
    std::vector<unsigned long long> laneWarmGen; // per lane: arena generation an un-captured full chunk has run with (0 = none)
    std::vector<uint8_t> viewBuf;         // pop_many's copy of a mini chunk's results (its place is free at once)
    long long graphLaunches = 0;          // chunks replayed from a graph (option "stream_graph_launches", read only)
    bool packed = false;                  // PS_FRAMES_PACKED: a frame's descriptors and points lie together, in the ring (ringDesc holds
    size_t packStride = 0;                // ringFrames x packStride bytes, ringPts is unused) as on the host: ONE upload per chunk
    std::vector<int32_t> nkRing; // row counts of the ring's slots (host-authoritative; every chunk uploads a snapshot)
    hipStream_t copyStream = nullptr;    // uploads
    hipStream_t copyOutStream = nullptr; // downloads
    std::vector<AsyncLane> lane;         // the lanes + ahead places, taken in ring order
    size_t offMask = 0, offPose = 0, offStats = 0, offNum = 0, resBytes = 0;
    int head = 0, tail = 0, inFlight = 0; // oldest chunk in flight, next lane to submit to
    // The block a pop hands to the caller changes hands instead of keeping its lane busy: the lane takes the spare pinned block
    // and is free at once (round 5's first form held the lane until the NEXT pop: of four lanes three computed).  Pops are in
    // order and a view lives until the next pop, so one block is held at any time.
    uint8_t *spareHres = nullptr, *spareHresDev = nullptr; // free pinned result block (+ its device view)
    uint8_t *heldHres = nullptr, *heldHresDev = nullptr;   // the block the caller's view points into (null: no view)
    int ringPos = 0;                      // slot the next frame goes to
    int prevPos = -1;                     // slot of the stream's latest frame (-1: the next frame has no predecessor)
    long long pairCounter = 0;
    int epoch = 0;
    int staged = 0;                       // frames collected in the next chunk's staging area by push_async
    std::vector<int32_t> stagedNk;
    int resultMode = 0;                   // PsStreamResults
    bool downloadsOnLane = false;         // downloads queued on the lane's own stream instead of the copy-out stream: when the
                                          // process has few hardware queues (GPU_MAX_HW_QUEUES < lanes + 6), or forced either
                                          // way by PUTSLAM_HIP_STREAM_DOWNLOADS_ON_LANE=0|1
    int cursor = 0;                       // ps_vo_stream_pop: next pair of the held view
    bool haveView = false;                // pop_many has handed out a block that has not been released yet
    PsHostPairResults view{};
    double dbgT[3] = {0, 0, 0};        // PUTSLAM_HIP_STREAM_DEBUG=1: host seconds inside the uploads' / the batched call's /
    long long dbgN = 0;                   // the downloads' submission, printed when the pipeline is released
};

namespace {

using psdev::CopySegs;
using psdev::ps_copy_segments;
using psdev::ps_pack_results_to_host;

int async_fail(PsVoStream *s, int code, const char *what, hipError_t e = hipSuccess) { return fail(s->ctx, code, what, e); }

#define PSA_HIP(call)                                                      \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) return async_fail(s, PS_ERR_HIP, #call, e_); \
    } while (0)

void async_drain(PsVoAsync *a)
{
    if (a->copyStream) (void)hipStreamSynchronize(a->copyStream);
    for (PsContext *c : a->laneCtx)
        if (c) (void)hipStreamSynchronize(c->stream);
    if (a->copyOutStream) (void)hipStreamSynchronize(a->copyOutStream);
}

void async_free(PsVoAsync *a)
{
    async_drain(a);
    if (a->dbgN > 0 && std::getenv("PUTSLAM_HIP_STREAM_DEBUG"))
        fprintf(stderr, "[ps stream] %lld chunks of <= %d frames on %d lanes (+ %d ahead): host us per chunk: upload %.1f, batched call %.1f, download %.1f\n",
                a->dbgN, a->B, a->lanes, a->ahead, 1e6 * a->dbgT[0] / a->dbgN, 1e6 * a->dbgT[1] / a->dbgN, 1e6 * a->dbgT[2] / a->dbgN);
    for (AsyncLane &l : a->lane) {
        release(l.meta);
        release(l.res);
        if (l.hmeta) (void)hipHostFree(l.hmeta);
        if (l.hres) (void)hipHostFree(l.hres);
        release(l.frames);
        if (l.gexec) (void)hipGraphExecDestroy(l.gexec);
        if (l.evRun) (void)hipEventDestroy(l.evRun);
        if (l.evDone) (void)hipEventDestroy(l.evDone);
    }
    for (PsContext *c : a->laneCtx)
        if (c) ps_context_destroy(c);
    if (a->loneHost) (void)hipHostFree(a->loneHost);
    if (a->loneReadEv) (void)hipEventDestroy(a->loneReadEv);
    if (a->spareHres) (void)hipHostFree(a->spareHres);
    if (a->heldHres) (void)hipHostFree(a->heldHres);
    for (hipEvent_t e : a->upEv)
        if (e) (void)hipEventDestroy(e);
    for (uint8_t *h : a->stagePool)
        if (h) (void)hipHostFree(h);
    release(a->ringDesc);
    release(a->ringPts);
    if (a->copyStream) (void)hipStreamDestroy(a->copyStream);
    if (a->copyOutStream) (void)hipStreamDestroy(a->copyOutStream);
    delete a;
}

bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // pageable memory: "invalid value", not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// chunks that can be accepted now: the free places (they are taken in ring order: free ones are contiguous from `tail`)
int async_room(const PsVoAsync *a)
{
    const int S = (int)a->lane.size();
    int freePlaces = 0;
    for (int i = 0; i < S && a->lane[(size_t)((a->tail + i) % S)].state == 0; ++i) ++freePlaces;
    return freePlaces;
}

// the pinned staging area of the chunk that will be uploaded next (chunk number chunkSeq): a chunk's area is used again
// lanes + ahead + 1 chunks later, and a chunk is only accepted while fewer than lanes + ahead chunks are alive (async_room) --
// by then the earlier user of the area has been popped, so its upload is long done
int stage_area(PsVoStream *s, uint8_t **out)
{
    PsVoAsync *a = s->async;
    uint8_t *&h = a->stagePool[(size_t)(a->chunkSeq % (long long)a->stagePool.size())];
    if (!h) {
        PSA_HIP(hipHostMalloc((void **)&h, a->packed ? (size_t)a->B * a->packStride : (size_t)a->B * s->cap * 44, hipHostMallocDefault));
        if (a->mini) PSA_HIP(hipHostGetDevicePointer((void **)&a->stagePoolDev[(size_t)(a->chunkSeq % (long long)a->stagePool.size())], h, 0));
    }
    *out = h;
    return PS_OK;
}

// One chunk on lane[tail] (free, checked by the caller): meta block, the batched call's launches behind the chunk's uploads,
// the download.
int async_launch(PsVoStream *s, const AsyncChunk &c)
{
    PsVoAsync *a = s->async;
    PsContext *ctx = s->ctx;
    const size_t cap = (size_t)s->cap;
    AsyncLane &l = a->lane[(size_t)a->tail];
    // consecutive chunks on consecutive lanes; a lane's stream orders the chunks it is given (its scratch arena is theirs in turn)
    // (launchSeq, the place's state, tail and inFlight are committed at the END of this function: a chunk that could not be
    // queued completely leaves the pipeline's bookkeeping as it found it -- async_abort_chunk does the rest)
    PsContext *lc = l.ctx = a->laneCtx[(size_t)(a->launchSeq % (long long)a->lanes)];
    a->diagLaunches++;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t1 = now();
    const int P = c.n - c.first;
    int32_t *pm = l.hmeta;
    for (int i = c.first; i < c.n; ++i) { // query = previous frame, train = current (matcher.cpp:470-471)
        pm[2 * (i - c.first)] = i == 0 ? c.prevPos : c.pos0 + i - 1;
        pm[2 * (i - c.first) + 1] = c.pos0 + i;
    }
    // (a snapshot of the ring's row counts as they are NOW: it may already hold those of later chunks, never other values for
    // the slots this chunk reads -- those are not written again while it is alive)
    memcpy(pm + 2 * (size_t)a->B, a->nkRing.data(), (size_t)a->ringFrames * sizeof(int32_t));
    CopySegs up{};
    up.src[0] = l.hmetaDev;
    up.dst[0] = l.meta.p;
    up.bytes[0] = ((unsigned long long)2 * a->B + a->ringFrames) * sizeof(int32_t);
    up.n = 1;
    hipLaunchKernelGGL(ps_copy_segments, dim3(1), dim3(256), 0, lc->stream, up); // (a few KB: beside the previous chunk's kernels)
    PS_HIP(hipGetLastError());
#ifdef PS_STREAM_DIAG
    // (experiment, results meaningless: the chunk's kernels do not wait for its upload -- what the upload -> lane dependency costs)
    static const bool diagNoUpWait = std::getenv("PUTSLAM_HIP_STREAM_DIAG_NO_UPWAIT") != nullptr;
    if (!diagNoUpWait)
#endif
    PS_HIP(hipStreamWaitEvent(lc->stream, c.up, 0));
    PsFrameSet fs;
    fs.desc = (const uint8_t *)a->ringDesc.p;
    fs.pts = a->packed ? (const float *)((const uint8_t *)a->ringDesc.p + cap * 32) : (const float *)a->ringPts.p;
    fs.nkpts = (const int32_t *)l.meta.p + 2 * (size_t)a->B;
    fs.numFrames = a->ringFrames;
    fs.maxKpts = s->cap;
    fs.descFrameStride = fs.ptsFrameStride = a->packed ? a->packStride : 0;
    uint8_t *dres = (uint8_t *)l.res.p;
    PsPairResults out;
    out.matches = (PsDMatch *)dres;
    out.inlierMask = dres + a->offMask;
    out.pose = (float *)(dres + a->offPose);
    out.stats = (PsRansacStats *)(dres + a->offStats);
    out.numMatches = (int32_t *)(dres + a->offNum);
    PsRansacConfig c2 = a->cfg;
    c2.seed = a->cfg.seed + (uint64_t)c.firstPair;
    int rc = PS_OK;
#ifdef PS_STREAM_DIAG
    // fault injection (tests/test_gpu_stream_async.py builds the library with -DPS_STREAM_DIAG; the shipped one does not contain
    // this): the batched call of the N-th chunk since configure fails before / after it has queued its launches
    static const char *diagFail = std::getenv("PUTSLAM_HIP_STREAM_DIAG_FAIL_CHUNK");
    static const char *diagFailAfter = std::getenv("PUTSLAM_HIP_STREAM_DIAG_FAIL_AFTER");
    if (diagFail && std::atoll(diagFail) == a->diagLaunches - 1) {
        lc->err = "injected failure (PS_STREAM_DIAG)";
        rc = PS_ERR_HIP;
    }
#endif
    if (rc == PS_OK) rc = ps_vo_pairs_device(lc, &a->prm, &c2, a->haveK ? a->K : nullptr, &fs, (const int32_t *)l.meta.p, P, &out);
    if (rc != PS_OK) {
        ctx->err = std::string("pipelined chunk: ") + lc->err;
        return rc;
    }
#ifdef PS_STREAM_DIAG
    if (diagFailAfter && std::atoll(diagFailAfter) == a->diagLaunches - 1)
        return fail(ctx, PS_ERR_HIP, "pipelined chunk: injected failure behind the batched call (PS_STREAM_DIAG)");
#endif
    const double t2 = now();
    // The download goes out on a stream of its own, behind an event, when the process has hardware queues to spare (async_build):
    // this runtime executes a device -> host hipMemcpyAsync as a blit kernel whatever stream it is queued on
    // (profiles/r05e/timeline_tail.txt), and on the lane's own stream that kernel sat between the lane's launches (29 % of the
    // time one was running, profiles/r05d); on the copy-out stream it overlaps the lane's NEXT chunk instead: + 11 ... 18 % with
    // six lanes.
    hipStream_t ds = a->downloadsOnLane ? lc->stream : a->copyOutStream;
    if (!a->downloadsOnLane) {
        PS_HIP(hipEventRecord(l.evRun, lc->stream));
        PS_HIP(hipStreamWaitEvent(ds, l.evRun, 0));
    }
    if (a->resultMode != PS_RESULTS_FULL) {
        // inliers / poses only: a kernel writes just those into the mapped pinned block, behind kernel 4 (on the download stream
        // too: on the lane's own stream the lane's next chunk waited behind its writes over the link)
        hipLaunchKernelGGL(ps_pack_results_to_host, dim3((unsigned)P), dim3(kBlock), 0, ds, (const PsDMatch *)dres,
                           (const int32_t *)(dres + a->offNum), (const uint8_t *)(dres + a->offMask), (const float *)(dres + a->offPose),
                           (const PsRansacStats *)(dres + a->offStats), s->cap, a->resultMode, (PsDMatch *)l.hresDev,
                           (float *)(l.hresDev + a->offPose), (PsRansacStats *)(l.hresDev + a->offStats),
                           (int32_t *)(l.hresDev + a->offNum));
        PS_HIP(hipGetLastError());
    }
    if (a->resultMode != PS_RESULTS_FULL) {
        // (written by ps_pack_results_to_host above)
    } else if (P == a->B) {
        PS_HIP(hipMemcpyAsync(l.hres, dres, a->resBytes, hipMemcpyDeviceToHost, ds));
    } else {
        const size_t p = (size_t)P;
        PS_HIP(hipMemcpyAsync(l.hres, dres, p * cap * sizeof(PsDMatch), hipMemcpyDeviceToHost, ds));
        PS_HIP(hipMemcpyAsync(l.hres + a->offMask, dres + a->offMask, p * cap, hipMemcpyDeviceToHost, ds));
        PS_HIP(hipMemcpyAsync(l.hres + a->offPose, dres + a->offPose, p * 64, hipMemcpyDeviceToHost, ds));
        PS_HIP(hipMemcpyAsync(l.hres + a->offStats, dres + a->offStats, p * sizeof(PsRansacStats), hipMemcpyDeviceToHost, ds));
        PS_HIP(hipMemcpyAsync(l.hres + a->offNum, dres + a->offNum, p * sizeof(int32_t), hipMemcpyDeviceToHost, ds));
    }
    PS_HIP(hipEventRecord(l.evDone, ds));
    const double t3 = now();
    if (const char *v = std::getenv("PUTSLAM_HIP_STREAM_DEBUG"))
        if (v[0] == '2')
            fprintf(stderr, "[chunk at pair %lld lane %d P %d] call %.1f download %.1f\n", c.firstPair, a->tail, P, 1e6 * (t2 - t1),
                    1e6 * (t3 - t2));
    a->dbgT[1] += t2 - t1;
    a->dbgT[2] += t3 - t2;
    a->dbgN++;
    l.state = 1;
    l.firstPair = c.firstPair;
    l.pairs = P;
    l.epoch = c.epoch;
    a->launchSeq++;
    a->tail = (a->tail + 1) % (int)a->lane.size();
    a->inFlight++;
    return PS_OK;
}

// A chunk could not be queued (an allocation or a HIP call failed somewhere between its upload and its download).  Whatever
// part of it WAS queued is drained -- nothing is in flight on a place that is still marked free --, the chunk's frames are
// dropped as a unit, and the stream continues as after ps_vo_stream_reset: the next frame has no predecessor, pair numbering
// restarts at 0, the epoch advances (include/putslam_hip.h).  Chunks submitted before keep their place, numbering and epoch.
int async_abort_chunk(PsVoStream *s, int rc)
{
    PsVoAsync *a = s->async;
    const std::string why = s->ctx->err;
    if (a->copyStream) (void)hipStreamSynchronize(a->copyStream);
    PsContext *lc = a->laneCtx.empty() ? nullptr : a->laneCtx[(size_t)((a->mini ? (long long)a->tail : a->launchSeq) % (long long)a->lanes)];
    if (lc) (void)hipStreamSynchronize(lc->stream);
    a->haloDev = nullptr;
    a->haloIsLone = false;
    if (a->copyOutStream) (void)hipStreamSynchronize(a->copyOutStream);
    (void)hipGetLastError();
    a->prevPos = -1;
    a->pairCounter = 0;
    a->epoch++;
    a->staged = 0;
    s->ctx->err = why + " -- pipelined stream: the chunk's frames were dropped; the next frame starts a new epoch (pair numbering from 0)";
    return rc;
}

// One chunk: n frames (pinned host memory: desc n x cap x 32, pts n x cap x 3; row counts nk) -> ring -> the next place, launched
// on the next lane's stream.  The caller has checked async_room().
int async_submit_body(PsVoStream *s, const uint8_t *desc, const float *pts, const int32_t *nk, int n);

using psdev::MiniMeta;
using psdev::ps_mini_copy_in;

// The launches of one mini chunk on its lane's stream: frames in, kernels 1 - 4, results out (what a place's graph holds).
int mini_enqueue(PsVoStream *s, AsyncLane &l, PsContext *lc, const Plan &pl, int P)
{
    PsVoAsync *a = s->async;
    PsContext *ctx = s->ctx;
    const size_t cap = (size_t)s->cap;
    MiniMeta *dm = (MiniMeta *)l.meta.p;
    const int groups = 8;
    unsigned copyGrid = (unsigned)((a->B + 1) * groups);
#ifdef PS_STREAM_DIAG
    // (experiment: what the lanes do when the frames cost nothing -- the meta block alone travels, the kernels run on whatever the
    // place's frames hold; results are meaningless)
    static const bool diagNoUpload = std::getenv("PUTSLAM_HIP_STREAM_DIAG_NO_UPLOAD") != nullptr;
    if (diagNoUpload && a->diagLaunches > 64) copyGrid = 0;
#endif
    if (copyGrid == 0) {
        CopySegs up{};
        up.src[0] = l.hmetaDev;
        up.dst[0] = dm;
        up.bytes[0] = sizeof(MiniMeta);
        up.n = 1;
        hipLaunchKernelGGL(ps_copy_segments, dim3(1), dim3(64), 0, lc->stream, up);
    } else
        hipLaunchKernelGGL(ps_mini_copy_in, dim3(copyGrid), dim3(256), 0, lc->stream, (const MiniMeta *)l.hmetaDev, dm,
                           (uint8_t *)l.frames.p, s->cap, (unsigned long long)a->packStride, groups);
    PS_HIP(hipGetLastError());
    PsFrameSet fs;
    fs.desc = (const uint8_t *)l.frames.p;
    fs.pts = (const float *)((const uint8_t *)l.frames.p + cap * 32);
    fs.nkpts = dm->nk;
    fs.numFrames = a->B + 1;
    fs.maxKpts = s->cap;
    fs.descFrameStride = fs.ptsFrameStride = a->packStride;
    uint8_t *dres = (uint8_t *)l.res.p;
    int rc = run_match_stage(lc, fs, dm->pairs, P, true, pl.pa, (PsDMatch *)dres, (int32_t *)(dres + a->offNum), 0);
    if (rc == PS_OK)
        rc = run_ransac_stage(lc, pl, P, s->cap, (const PsDMatch *)dres, (const int32_t *)(dres + a->offNum), s->cap, (float *)(dres + a->offPose),
                              dres + a->offMask, (PsRansacStats *)(dres + a->offStats), 2);
    if (rc != PS_OK) {
        ctx->err = std::string("pipelined chunk: ") + lc->err;
        return rc;
    }
    if (a->resultMode != PS_RESULTS_FULL) {
        hipLaunchKernelGGL(ps_pack_results_to_host, dim3((unsigned)P), dim3(kBlock), 0, lc->stream, (const PsDMatch *)dres,
                           (const int32_t *)(dres + a->offNum), (const uint8_t *)(dres + a->offMask), (const float *)(dres + a->offPose),
                           (const PsRansacStats *)(dres + a->offStats), s->cap, a->resultMode, (PsDMatch *)l.hresDev,
                           (float *)(l.hresDev + a->offPose), (PsRansacStats *)(l.hresDev + a->offStats), (int32_t *)(l.hresDev + a->offNum));
    } else {
        const size_t p = (size_t)P;
        CopySegs down{};
        const size_t off[5] = {0, a->offMask, a->offPose, a->offStats, a->offNum};
        const size_t len[5] = {p * cap * sizeof(PsDMatch), (p * cap + 3) & ~(size_t)3, p * 64, p * sizeof(PsRansacStats), p * sizeof(int32_t)};
        for (int k = 0; k < 5; ++k) {
            down.src[k] = dres + off[k];
            down.dst[k] = l.hresDev + off[k];
            down.bytes[k] = len[k];
        }
        down.n = 5;
        const unsigned groupsOut = (unsigned)((len[0] / 16 + 255) / 256);
        hipLaunchKernelGGL(ps_copy_segments, dim3(groupsOut < 1 ? 1 : (groupsOut > 32 ? 32 : groupsOut)), dim3(256), 0, lc->stream, down);
    }
    PS_HIP(hipGetLastError());
    return PS_OK;
}

// One mini chunk: the n frames staged in `h` (packed layout) -> the next place, on that place's lane.  The caller has checked
// async_room().
int mini_submit_body(PsVoStream *s, uint8_t *h, const int32_t *nk, int n)
{
    PsVoAsync *a = s->async;
    PsContext *ctx = s->ctx;
    // the chunk's frames as the copy-in kernel sees them: the staging area they were collected in, or -- pinned packed frames
    // handed over by ps_vo_stream_push_many_packed (PsVoAsync::inPlaceDev) -- the caller's own memory, read in place
    const bool inPlace = a->inPlaceDev != nullptr;
    const uint8_t *hDev = inPlace ? a->inPlaceDev : a->stagePoolDev[(size_t)(a->chunkSeq % (long long)a->stagePool.size())];
    a->inPlaceDev = nullptr;
    const int first = a->prevPos >= 0 ? 0 : 1;
    const int P = n - first;
    const uint8_t *haloWas = a->haloDev;
    const int haloNkWas = a->haloNk;
    const bool haloWasLone = a->haloIsLone;
    if (P <= 0) {
        // A lone first frame (the stream's first, or the first after a reset): nothing to run, no place taken -- and therefore no
        // staging area either: the areas rotate with the LAUNCHED chunks, which is what makes their reuse safe (an area is written
        // again lanes + ahead + 1 launched chunks later: by then its chunk AND the successor that reads its last frame as the
        // previous one have been popped).  Round 6's first form let such a frame keep its area and advance the rotation: every
        // reset moved the reuse one chunk closer, until a new frame could be staged over a frame an in-flight chunk had not read
        // yet (found by the stream fuzz: 17 of 15 000 configurations, all with a reset and chunks of one or two frames).  The frame
        // moves to a buffer of its own instead; that buffer is written again only after the chunk that read it has finished.
        if (inPlace) {
            a->haloDev = hDev; // (the caller's pinned memory: untouched until the pair it is the previous frame of has been popped)
            a->haloIsLone = false;
        } else {
            if (!a->loneHost) {
                PS_HIP(hipHostMalloc((void **)&a->loneHost, a->packStride, hipHostMallocDefault));
                PS_HIP(hipHostGetDevicePointer((void **)&a->loneDev, a->loneHost, 0));
                PS_HIP(hipEventCreateWithFlags(&a->loneReadEv, hipEventDisableTiming));
            }
            if (a->loneReadPending) {
                PS_HIP(hipEventSynchronize(a->loneReadEv));
                a->loneReadPending = false;
            }
            const size_t cap = (size_t)s->cap;
            memcpy(a->loneHost, h, (size_t)nk[0] * 32);
            memcpy(a->loneHost + cap * 32, h + cap * 32, (size_t)nk[0] * 12);
            a->haloDev = a->loneDev;
            a->haloIsLone = true;
        }
        a->haloNk = nk[0];
        a->prevPos = 0;
        return PS_OK;
    }
    a->chunkSeq++;
    // (the stream's latest frame from now on: the next chunk reads it from this staging slot -- or from the caller's pinned block)
    a->haloDev = hDev + (size_t)(n - 1) * a->packStride;
    a->haloNk = nk[n - 1];
    a->haloIsLone = false;
    a->prevPos = 0;
    const int place = a->tail;
    AsyncLane &l = a->lane[(size_t)place];
    const int laneIdx = place % a->lanes; // (fixed per place: the place's graph holds its lane's arena pointers)
    PsContext *lc = l.ctx = a->laneCtx[(size_t)laneIdx];
    a->diagLaunches++;
    MiniMeta *hm = (MiniMeta *)l.hmeta;
    memset(hm, 0, sizeof *hm);
    for (int i = 0; i < P; ++i) { // query = previous frame, train = current (matcher.cpp:470-471)
        hm->pairs[2 * i] = first + i;
        hm->pairs[2 * i + 1] = first + i + 1;
    }
    hm->nk[0] = first ? 0 : haloNkWas;
    for (int i = 0; i < n; ++i) hm->nk[1 + i] = nk[i];
    hm->n = n;
    hm->first = first;
    hm->seed = a->cfg.seed + (uint64_t)a->pairCounter;
    hm->src[0] = first ? nullptr : haloWas;
    for (int i = 0; i < n; ++i) hm->src[1 + i] = hDev + (size_t)i * a->packStride;
    int rc = PS_OK;
#ifdef PS_STREAM_DIAG
    static const char *diagFail = std::getenv("PUTSLAM_HIP_STREAM_DIAG_FAIL_CHUNK");
    static const char *diagFailAfter = std::getenv("PUTSLAM_HIP_STREAM_DIAG_FAIL_AFTER");
    if (diagFail && std::atoll(diagFail) == a->diagLaunches - 1) return fail(ctx, PS_ERR_HIP, "pipelined chunk: injected failure (PS_STREAM_DIAG)");
#endif
    const bool full = P == a->B && !first;
    bool launched = false;
    if (full && a->miniGraphs && s->graphsEnabled) { // (PUTSLAM_HIP_STREAM_GRAPH=1; PUTSLAM_HIP_NO_GRAPH=1 at ps_vo_stream_create wins)
        if (l.gexec && l.gexecGen != lc->arenaGen) { // the lane's arena has moved since the capture
            (void)hipGraphExecDestroy(l.gexec);
            l.gexec = nullptr;
        }
        if (!l.gexec && a->laneWarmGen[(size_t)laneIdx] == lc->arenaGen && lc->arenaGen != 0) {
            // an un-captured full chunk has sized this lane's arena: capture the place's chunk
            Plan pl;
            rc = make_plan(lc, &a->prm, &a->cfg, a->haveK ? a->K : nullptr, s->cap, s->cap, pl);
            if (rc == PS_OK) {
                pl.ma.seedDev = reinterpret_cast<const uint64_t *>(&((MiniMeta *)l.meta.p)->seed);
                rc = prepare_score(lc, pl, P, s->cap);
            }
            if (rc == PS_OK) {
                PS_ENSURE(lc->keys, (size_t)P * s->cap * sizeof(uint32_t));
                rc = keys_clean(lc, (size_t)P * s->cap * sizeof(uint32_t)); // (outside the capture: a replay finds the block as its capture did)
            }
            if (rc != PS_OK) {
                ctx->err = std::string("pipelined chunk: ") + lc->err;
                return rc;
            }
            if (lc->arenaGen == a->laneWarmGen[(size_t)laneIdx]) {
                hipGraph_t graph = nullptr;
                if (hipStreamBeginCapture(lc->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const int r = mini_enqueue(s, l, lc, pl, P);
                    const hipError_t e2 = hipStreamEndCapture(lc->stream, &graph);
                    if (r == PS_OK && e2 == hipSuccess && graph && hipGraphInstantiate(&l.gexec, graph, nullptr, nullptr, 0) != hipSuccess) l.gexec = nullptr;
                    if (r != PS_OK || e2 != hipSuccess) l.gexec = nullptr;
                    if (graph) (void)hipGraphDestroy(graph);
                }
                if (!l.gexec) {
                    s->graphsEnabled = false; // capture is not available here: ordinary launches from now on
                    (void)hipGetLastError();
                    ctx->err.clear();
                } else {
                    l.gexecGen = lc->arenaGen;
                }
            }
        }
        if (l.gexec) {
            PS_HIP(hipGraphLaunch(l.gexec, lc->stream));
            a->graphLaunches++;
            launched = true;
        }
    }
    if (!launched) {
        // the plan of a full chunk is made once per place (parameters, P and the lane's arena do not change between chunks): the
        // host side of a chunk is its staging copy and seven launches
        const bool cached = full && l.planGen != 0 && l.planGen == lc->arenaGen;
        Plan local;
        Plan &pl = full ? l.plan : local;
        if (!cached) {
            l.planGen = 0;
            rc = make_plan(lc, &a->prm, &a->cfg, a->haveK ? a->K : nullptr, s->cap, s->cap, pl);
            if (rc == PS_OK) {
                pl.ma.seedDev = reinterpret_cast<const uint64_t *>(&((MiniMeta *)l.meta.p)->seed);
                rc = prepare_score(lc, pl, P, s->cap);
            }
            if (rc != PS_OK) {
                ctx->err = std::string("pipelined chunk: ") + lc->err;
                return rc;
            }
        }
        rc = mini_enqueue(s, l, lc, pl, P);
        if (rc != PS_OK) return rc;
        if (full) {
            a->laneWarmGen[(size_t)laneIdx] = lc->arenaGen;
            l.planGen = lc->arenaGen; // (the plan points at the blocks prepare_score sized -- counts, parked models, stop tables --;
                                      // the launches above only grow OTHER blocks, so the plan stays good for the arena as it is now)
        }
    }
#ifdef PS_STREAM_DIAG
    if (diagFailAfter && std::atoll(diagFailAfter) == a->diagLaunches - 1)
        return fail(ctx, PS_ERR_HIP, "pipelined chunk: injected failure behind the batched call (PS_STREAM_DIAG)");
#endif
    PS_HIP(hipEventRecord(l.evDone, lc->stream));
    if (haloWasLone && !first) { // this chunk read the lone-frame buffer as its previous frame: the buffer is free again behind it
        PS_HIP(hipEventRecord(a->loneReadEv, lc->stream));
        a->loneReadPending = true;
    }
    a->dbgN++;
    l.state = 1;
    l.firstPair = a->pairCounter;
    l.pairs = P;
    l.epoch = a->epoch;
    a->pairCounter += P;
    a->launchSeq++;
    a->tail = (a->tail + 1) % (int)a->lane.size();
    a->inFlight++;
    return PS_OK;
}

int async_submit(PsVoStream *s, const uint8_t *desc, const float *pts, const int32_t *nk, int n)
{
    const int rc = s->async->mini ? mini_submit_body(s, const_cast<uint8_t *>(desc), nk, n) : async_submit_body(s, desc, pts, nk, n);
    return rc == PS_OK ? PS_OK : async_abort_chunk(s, rc);
}

int async_submit_body(PsVoStream *s, const uint8_t *desc, const float *pts, const int32_t *nk, int n)
{
    PsVoAsync *a = s->async;
    PsContext *ctx = s->ctx;
    const size_t cap = (size_t)s->cap;
    const int pos0 = (a->ringPos + n <= a->ringFrames) ? a->ringPos : 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    hipEvent_t ev = a->upEv[(size_t)(a->chunkSeq % (long long)a->upEv.size())];
    a->chunkSeq++;
    // (a build with -DPS_STREAM_DIAG and PUTSLAM_HIP_STREAM_DIAG_NO_UPLOAD=1 skips the uploads after the first 64 chunks -- the chunks
    // then run on whatever frames the ring holds and the results are meaningless; it told what the pipeline does when the link
    // costs nothing, profiles/r05h/stream_limits.txt.  The shipped library does not contain it.)
#ifdef PS_STREAM_DIAG
    static const bool diagNoUpload = std::getenv("PUTSLAM_HIP_STREAM_DIAG_NO_UPLOAD") != nullptr;
    if (!(diagNoUpload && a->chunkSeq > 64))
#endif
    {
        if (a->packed) { // (`desc` = the chunk's packed frames: ONE transfer, no idle link between a descriptor and a point upload)
            PS_HIP(hipMemcpyAsync((uint8_t *)a->ringDesc.p + (size_t)pos0 * a->packStride, desc, (size_t)n * a->packStride,
                                  hipMemcpyHostToDevice, a->copyStream));
        } else {
            PS_HIP(hipMemcpyAsync((uint8_t *)a->ringDesc.p + (size_t)pos0 * cap * 32, desc, (size_t)n * cap * 32, hipMemcpyHostToDevice,
                                  a->copyStream));
            PS_HIP(hipMemcpyAsync((uint8_t *)a->ringPts.p + (size_t)pos0 * cap * 12, pts, (size_t)n * cap * 12, hipMemcpyHostToDevice,
                                  a->copyStream));
        }
    }
    PS_HIP(hipEventRecord(ev, a->copyStream));
    for (int i = 0; i < n; ++i) a->nkRing[(size_t)(pos0 + i)] = nk[i];
    AsyncChunk c;
    c.pos0 = pos0;
    c.n = n;
    c.first = a->prevPos >= 0 ? 0 : 1; // the stream's first frame has no predecessor (matcher.cpp:17-64)
    c.prevPos = a->prevPos;
    c.firstPair = a->pairCounter;
    c.epoch = a->epoch;
    c.up = ev;
    const int P = n - c.first;
    a->ringPos = pos0 + n;
    a->prevPos = pos0 + n - 1;
    a->dbgT[0] += now() - t0;
    if (P <= 0) {
        // a lone first frame: nothing to run, no lane taken; its staging area (if it came through one) is reused by the
        // next push, so the upload is waited for here -- and the areas' rotation does not advance: it follows the LAUNCHED chunks
        // (an area is written again lanes + ahead + 1 launched chunks later, when its own chunk has been popped)
        PS_HIP(hipStreamSynchronize(a->copyStream));
        a->chunkSeq--;
        return PS_OK;
    }
    a->pairCounter += P;
    return async_launch(s, c);
}

int async_submit_staged(PsVoStream *s)
{
    PsVoAsync *a = s->async;
    if (a->staged == 0) return PS_OK;
    const size_t cap = (size_t)s->cap;
    const int n = a->staged;
    // (the staging area is the one push_async has been filling: that of chunk number chunkSeq)
    uint8_t *h = nullptr;
    int rc = stage_area(s, &h);
    if (rc) return rc;
    a->staged = 0; // (whatever happens below, these frames are not submitted a second time)
    return async_submit(s, h, a->packed ? nullptr : reinterpret_cast<const float *>(h + (size_t)a->B * cap * 32), a->stagedNk.data(), n);
}

// device / pinned blocks, streams, events and lane contexts of a freshly configured pipeline
int async_build(PsVoStream *s)
{
    PsVoAsync *a = s->async;
    PsContext *ctx = s->ctx;
    const size_t cap = (size_t)s->cap, B = (size_t)a->B;
    a->offMask = B * cap * sizeof(PsDMatch);
    a->offPose = a->offMask + ((B * cap + 63) & ~(size_t)63);
    a->offStats = a->offPose + B * 64;
    a->offNum = a->offStats + ((B * sizeof(PsRansacStats) + 63) & ~(size_t)63);
    a->resBytes = a->offNum + B * sizeof(int32_t);
    a->nkRing.assign((size_t)a->ringFrames, 0);
    a->stagedNk.assign(B, 0);
    a->upEv.assign((size_t)(a->lanes + a->ahead + 1), nullptr);
    a->stagePool.assign((size_t)(a->lanes + a->ahead + 1), nullptr);
    for (hipEvent_t &e : a->upEv) PS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    a->packed = s->asyncFrameLayout == PS_FRAMES_PACKED;
    a->packStride = (cap * 44 + 15) & ~(size_t)15;
    {
        // small chunks as one graph per place (PsVoAsync::mini); PUTSLAM_HIP_STREAM_MINI=0 keeps round 5's form for them (A/B, tests)
        const char *m = std::getenv("PUTSLAM_HIP_STREAM_MINI");
        a->mini = a->B <= psdev::kMiniFrames && !(m && std::atoi(m) == 0);
        const char *g = std::getenv("PUTSLAM_HIP_STREAM_GRAPH");
        a->miniGraphs = g && std::atoi(g) != 0;
    }
    if (a->mini) {
        a->packed = true; // (the staging areas: one block per frame, as the copy-in kernel reads them)
        a->laneWarmGen.assign((size_t)a->lanes, 0);
        a->stagePoolDev.assign(a->stagePool.size(), nullptr);
        a->viewBuf.assign((a->resBytes + 15) & ~(size_t)15, 0);
    } else if (a->packed) {
        PS_ENSURE(a->ringDesc, (size_t)a->ringFrames * a->packStride);
    } else {
        PS_ENSURE(a->ringDesc, (size_t)a->ringFrames * cap * 32);
        PS_ENSURE(a->ringPts, (size_t)a->ringFrames * cap * 12);
    }
    PS_HIP(hipStreamCreateWithFlags(&a->copyStream, hipStreamNonBlocking));
    PS_HIP(hipStreamCreateWithFlags(&a->copyOutStream, hipStreamNonBlocking));
    // A stream of its own for the downloads pays only when it also gets a hardware queue of its own: with the runtime's
    // default of four queues (or eight shared with the host's other streams) it lands on a lane's or the upload stream's queue
    // and serialises with it -- 130 k instead of 390 k frame-pairs/s in bench.py's process (profiles/r05f/hw_queues.txt).  The
    // runtime has no query for the number of queues; the environment variable that sets it is read instead.
    {
        int queues = 4; // (ROCclr's default)
        if (const char *q = std::getenv("GPU_MAX_HW_QUEUES")) queues = std::atoi(q);
        a->downloadsOnLane = queues < a->lanes + 6;
    }
    if (const char *v = std::getenv("PUTSLAM_HIP_STREAM_DOWNLOADS_ON_LANE")) a->downloadsOnLane = std::atoi(v) != 0;
    PS_HIP(hipHostMalloc((void **)&a->spareHres, (a->resBytes + 15) & ~(size_t)15, hipHostMallocDefault));
    PS_HIP(hipHostGetDevicePointer((void **)&a->spareHresDev, a->spareHres, 0));
    a->laneCtx.assign((size_t)a->lanes, nullptr);
    for (PsContext *&c : a->laneCtx) {
        int rc = ps_context_create(ctx->device, &c);
        if (rc != PS_OK) {
            c = nullptr;
            return fail(ctx, rc, "ps_vo_stream_configure_async: lane context");
        }
        psi_copy_options(c, ctx); // the lanes run what the stream's context would
        // Chunks on different lanes overlap, so the staged scoring pays from smaller batches on (prepare_score) -- from chunks of
        // 48 frames on: below, a chunk's extra launches cost more than its abandoned evaluations save (chunks of 32 / 64 frames on
        // three lanes, E1 / fixed / H = 4096: complete 204 k / 251 k, staged 156 k / 301 k frame-pairs/s, profiles/r06u/stream_small_chunks.txt)
        // (the decision is the stream's own: below 48 frames per chunk the lanes run as lone contexts whatever the stream's context says)
        if (B < 48) c->sideBySide = 0;
        else if (a->lanes > 1 && c->sideBySide < a->lanes) c->sideBySide = a->lanes;
    }
    a->lane.resize((size_t)(a->lanes + a->ahead));
    const size_t metaBytes = std::max(((size_t)2 * B + a->ringFrames) * sizeof(int32_t), sizeof(psdev::MiniMeta));
    for (AsyncLane &l : a->lane) {
        if (a->mini) PS_ENSURE(l.frames, (B + 1) * a->packStride);
        PS_ENSURE(l.meta, metaBytes);
        PS_ENSURE(l.res, (a->resBytes + 15) & ~(size_t)15);
        PS_HIP(hipHostMalloc((void **)&l.hmeta, metaBytes, hipHostMallocDefault));
        PS_HIP(hipHostMalloc((void **)&l.hres, (a->resBytes + 15) & ~(size_t)15, hipHostMallocDefault));
        PS_HIP(hipHostGetDevicePointer((void **)&l.hmetaDev, l.hmeta, 0));
        PS_HIP(hipHostGetDevicePointer((void **)&l.hresDev, l.hres, 0));
        PS_HIP(hipEventCreateWithFlags(&l.evRun, hipEventDisableTiming));
        PS_HIP(hipEventCreateWithFlags(&l.evDone, hipEventDisableTiming));
    }
    return PS_OK;
}

void release_view(PsVoAsync *a)
{
    if (a->heldHres) { // the block goes back to being the spare one
        a->spareHres = a->heldHres;
        a->spareHresDev = a->heldHresDev;
        a->heldHres = a->heldHresDev = nullptr;
    }
    a->cursor = 0;
    a->haveView = false;
    memset(&a->view, 0, sizeof a->view);
}

// Several frames into a stream of mini chunks (push_many / push_many_packed), with push_many's contract: frames staged by
// push_async before go first, as a chunk of their own; these frames follow in chunks of at most chunkFrames, the last one
// included -- everything is submitted when the call returns.  All or nothing: a place for every one of those chunks.
int mini_push_frames(PsVoStream *s, const uint8_t *desc, size_t descStride, const uint8_t *pts, size_t ptsStride, const int32_t *nkpts, int numFrames)
{
    PsVoAsync *a = s->async;
    const size_t cap = (size_t)s->cap;
    const int need = (a->staged > 0 ? 1 : 0) + (numFrames + a->B - 1) / a->B;
    if (need > async_room(a)) return async_fail(s, PS_ERR_BUSY, "ps_vo_stream_push_many: not enough room for these frames (pop results first, or push fewer)");
    int rc = async_submit_staged(s);
    if (rc) return rc;
    // Pinned packed frames are read in place by the copy-in kernel (88 KB less for the host to copy per 2000-keypoint frame): as
    // with the ring form they stay untouched until the results of their frames -- for a chunk's last frame: of the NEXT pair,
    // which reads it as its previous frame -- have been popped.
    if (descStride == a->packStride && pts == desc + cap * 32 && is_pinned_host(desc)) {
        uint8_t *dev = nullptr;
        if (hipHostGetDevicePointer((void **)&dev, const_cast<uint8_t *>(desc), 0) == hipSuccess && dev) {
            for (int f0 = 0; f0 < numFrames; f0 += a->B) {
                const int n = numFrames - f0 < a->B ? numFrames - f0 : a->B;
                a->inPlaceDev = dev + (size_t)f0 * a->packStride;
                rc = async_submit(s, desc + (size_t)f0 * a->packStride, nullptr, nkpts + f0, n);
                a->inPlaceDev = nullptr;
                if (rc) return rc;
            }
            return PS_OK;
        }
        (void)hipGetLastError();
    }
    for (int f = 0; f < numFrames; ++f) {
        uint8_t *h = nullptr;
        rc = stage_area(s, &h);
        if (rc) return rc;
        uint8_t *hd = h + (size_t)a->staged * a->packStride;
        memcpy(hd, desc + (size_t)f * descStride, (size_t)nkpts[f] * 32);
        memcpy(hd + cap * 32, pts + (size_t)f * ptsStride, (size_t)nkpts[f] * 12);
        a->stagedNk[(size_t)a->staged] = nkpts[f];
        a->staged++;
        if (a->staged == a->B || f == numFrames - 1) {
            rc = async_submit_staged(s);
            if (rc) return rc;
        }
    }
    return PS_OK;
}

} // namespace

static void async_release(PsVoStream *s)
{
    if (!s || !s->async) return;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    async_free(s->async);
    s->async = nullptr;
}

static int async_reset(PsVoStream *s)
{
    PsVoAsync *a = s->async;
    int rc = bind(s->ctx);
    if (rc) return rc;
    if (a->staged > 0) { // (its place was reserved when push_async took the chunk's first frame)
        rc = async_submit_staged(s);
        if (rc) return rc;
    }
    a->prevPos = -1;
    a->pairCounter = 0;
    a->epoch++;
    return PS_OK;
}

extern "C" {

int ps_vo_stream_configure_async(PsVoStream *s, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                                 int chunkFrames, int lanes)
{
    if (!s) return PS_ERR_BAD_ARG;
    PsContext *ctx = s->ctx;
    int rc = bind(ctx);
    if (rc) return rc;
    if (!params || !cfg) return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_configure_async: null params/config");
    if (cfg->sampleIdx) return fail(ctx, PS_ERR_BAD_ARG, "explicit sample streams are not supported by the streaming calls");
    if (chunkFrames == 0) chunkFrames = 128;
    // Lanes and places by chunk size, measured (profiles/r06i/stream_lanes_places.txt; round 5 ran six places in all, as six or
    // three lanes): chunks of a hundred frames and more fill the chip by themselves and run best as TWO launch chains with ONE more
    // chunk queued behind them -- 417 / 504 / 550 k frame-pairs/s at 125 / 250 / 500 frames per chunk against 398 / 474 / 445 k
    // with six places -- the same finding as the batch queue's (whole batches to two chains in turn); chunks of 48 .. 95 frames two
    // lanes and four places; smaller ones three lanes and six places; one to four frames six lanes (PsVoAsync::mini).
    if (lanes == 0) lanes = chunkFrames >= 48 ? 2 : (chunkFrames > psdev::kMiniFrames ? 3 : 6);
    if (chunkFrames < 1 || chunkFrames > 1024 || lanes < 2 || lanes > 8)
        return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_configure_async: chunkFrames 1..1024, lanes 2..8");
    {
        // the parameters are checked here, not at the first chunk: a plan on the stream's own context
        Plan pl;
        rc = make_plan(ctx, params, cfg, K, s->cap, s->cap, pl);
        if (rc) return rc;
    }
    if (s->async) { // re-configuration: drain, drop what has not been popped
        async_free(s->async);
        s->async = nullptr;
    }
    PsVoAsync *a = new PsVoAsync();
    s->async = a;
    a->B = chunkFrames;
    a->lanes = lanes;
    a->ahead = ctx->streamAhead >= 0 ? ctx->streamAhead : (chunkFrames >= 96 ? 1 : (chunkFrames >= 48 ? 2 : (lanes < 6 ? 6 - lanes : 0)));
    a->ringFrames = (lanes + a->ahead + 2) * chunkFrames;
    a->prm = *params;
    a->cfg = *cfg;
    a->cfg.sampleIdx = nullptr;
    a->haveK = K != nullptr;
    if (K) memcpy(a->K, K, sizeof a->K);
    a->resultMode = s->asyncResultMode;
    rc = async_build(s);
    if (rc != PS_OK) { // (the error text stays in the stream's context)
        const std::string why = ctx->err;
        async_free(a);
        s->async = nullptr;
        ctx->err = why;
    }
    return rc;
}

int ps_vo_stream_set_result_mode(PsVoStream *s, int mode)
{
    if (!s) return PS_ERR_BAD_ARG;
    if (mode < PS_RESULTS_FULL || mode > PS_RESULTS_POSES) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_set_result_mode: 0 (full), 1 (inliers) or 2 (poses)");
    s->asyncResultMode = mode;
    return PS_OK;
}

int ps_vo_stream_push_async(PsVoStream *s, const uint8_t *desc, size_t descStep, const float *pts, int n)
{
    if (!s) return PS_ERR_BAD_ARG;
    int rc = bind(s->ctx);
    if (rc) return rc;
    PsVoAsync *a = s->async;
    if (!a) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_async: call ps_vo_stream_configure_async first");
    if (n < 0 || n > s->cap || (n > 0 && (!desc || !pts)) || descStep < PS_DESC_BYTES)
        return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_async: bad argument");
    // (the first frame of a chunk reserves the chunk's place; the later ones ride on it)
    if (a->staged == 0 && async_room(a) < 1) return async_fail(s, PS_ERR_BUSY, "ps_vo_stream_push_async: no room (pop results first)");
    uint8_t *h = nullptr;
    rc = stage_area(s, &h);
    if (rc) return rc;
    const size_t cap = (size_t)s->cap;
    uint8_t *hd = a->packed ? h + (size_t)a->staged * a->packStride : h + (size_t)a->staged * cap * 32;
    uint8_t *hp = a->packed ? hd + cap * 32 : h + (size_t)a->B * cap * 32 + (size_t)a->staged * cap * 12;
    if (descStep == PS_DESC_BYTES) {
        if (n > 0) memcpy(hd, desc, (size_t)n * 32);
    } else {
        for (int i = 0; i < n; ++i) memcpy(hd + (size_t)i * 32, desc + (size_t)i * descStep, 32);
    }
    if (n > 0) memcpy(hp, pts, (size_t)n * 12);
    a->stagedNk[(size_t)a->staged] = n;
    a->staged++;
    if (a->staged == a->B) return async_submit_staged(s);
    return PS_OK;
}

int ps_vo_stream_flush(PsVoStream *s)
{
    if (!s) return PS_ERR_BAD_ARG;
    int rc = bind(s->ctx);
    if (rc) return rc;
    PsVoAsync *a = s->async;
    if (!a) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_flush: call ps_vo_stream_configure_async first");
    return async_submit_staged(s); // (the partly filled chunk's place was reserved by its first frame)
}

int ps_vo_stream_push_many(PsVoStream *s, const uint8_t *desc, const float *pts, const int32_t *nkpts, int numFrames)
{
    if (!s) return PS_ERR_BAD_ARG;
    int rc = bind(s->ctx);
    if (rc) return rc;
    PsVoAsync *a = s->async;
    if (!a) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_many: call ps_vo_stream_configure_async first");
    if (numFrames < 0 || (numFrames > 0 && (!desc || !pts || !nkpts)))
        return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_many: bad argument");
    for (int i = 0; i < numFrames; ++i)
        if (nkpts[i] < 0 || nkpts[i] > s->cap) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_many: row count out of range");
    if (numFrames == 0) return PS_OK;
    if (a->mini) return mini_push_frames(s, desc, (size_t)s->cap * 32, reinterpret_cast<const uint8_t *>(pts), (size_t)s->cap * 12, nkpts, numFrames);
    // places needed: one for the frames push_async has staged, one per chunk of these frames -- all or nothing
    const int chunks = (numFrames + a->B - 1) / a->B;
    const int need = chunks + (a->staged > 0 ? 1 : 0);
    if (need > async_room(a))
        return async_fail(s, PS_ERR_BUSY, "ps_vo_stream_push_many: not enough room for these frames (pop results first, or push fewer)");
    rc = async_submit_staged(s);
    if (rc) return rc;
    const size_t cap = (size_t)s->cap;
    // pinned frames are uploaded in place; pageable ones go through the lane's pinned staging area first
    const bool inPlace = !a->packed && is_pinned_host(desc) && is_pinned_host(pts);
    for (int f0 = 0; f0 < numFrames; f0 += a->B) {
        const int n = numFrames - f0 < a->B ? numFrames - f0 : a->B;
        const uint8_t *d = desc + (size_t)f0 * cap * 32;
        const float *p = pts + (size_t)f0 * cap * 3;
        if (a->packed) { // two host arrays into a packed ring: through the staging area, frame by frame
            uint8_t *h = nullptr;
            rc = stage_area(s, &h);
            if (rc) return rc;
            for (int i = 0; i < n; ++i) {
                memcpy(h + (size_t)i * a->packStride, d + (size_t)i * cap * 32, (size_t)nkpts[f0 + i] * 32);
                memcpy(h + (size_t)i * a->packStride + cap * 32, p + (size_t)i * cap * 3, (size_t)nkpts[f0 + i] * 12);
            }
            d = h;
            p = nullptr;
        } else if (!inPlace) {
            uint8_t *h = nullptr;
            rc = stage_area(s, &h);
            if (rc) return rc;
            memcpy(h, d, (size_t)n * cap * 32);
            memcpy(h + (size_t)a->B * cap * 32, p, (size_t)n * cap * 12);
            d = h;
            p = reinterpret_cast<const float *>(h + (size_t)a->B * cap * 32);
        }
        rc = async_submit(s, d, p, nkpts + f0, n);
        if (rc) return rc;
    }
    return PS_OK;
}

int ps_vo_stream_set_frame_layout(PsVoStream *s, int layout)
{
    if (!s) return PS_ERR_BAD_ARG;
    if (layout != PS_FRAMES_TWO_ARRAYS && layout != PS_FRAMES_PACKED)
        return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_set_frame_layout: 0 (two arrays) or 1 (packed frames)");
    s->asyncFrameLayout = layout;
    return PS_OK;
}

size_t ps_vo_stream_packed_stride(const PsVoStream *s) { return s ? (((size_t)s->cap * 44 + 15) & ~(size_t)15) : 0; }

int ps_vo_stream_push_many_packed(PsVoStream *s, const uint8_t *frames, size_t frameStride, const int32_t *nkpts, int numFrames)
{
    if (!s) return PS_ERR_BAD_ARG;
    int rc = bind(s->ctx);
    if (rc) return rc;
    PsVoAsync *a = s->async;
    if (!a) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_many_packed: call ps_vo_stream_configure_async first");
    if (!a->packed)
        return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_many_packed: the stream was configured for two arrays (ps_vo_stream_set_frame_layout "
                                             "before ps_vo_stream_configure_async)");
    if (numFrames < 0 || (numFrames > 0 && (!frames || !nkpts)) || frameStride != a->packStride)
        return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_many_packed: bad argument (frameStride must be ps_vo_stream_packed_stride())");
    for (int i = 0; i < numFrames; ++i)
        if (nkpts[i] < 0 || nkpts[i] > s->cap) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_push_many_packed: row count out of range");
    if (numFrames == 0) return PS_OK;
    if (a->mini) return mini_push_frames(s, frames, a->packStride, frames + (size_t)s->cap * 32, a->packStride, nkpts, numFrames);
    const int chunks = (numFrames + a->B - 1) / a->B;
    const int need = chunks + (a->staged > 0 ? 1 : 0);
    if (need > async_room(a))
        return async_fail(s, PS_ERR_BUSY, "ps_vo_stream_push_many_packed: not enough room for these frames (pop results first, or push fewer)");
    rc = async_submit_staged(s);
    if (rc) return rc;
    const bool inPlace = is_pinned_host(frames);
    for (int f0 = 0; f0 < numFrames; f0 += a->B) {
        const int n = numFrames - f0 < a->B ? numFrames - f0 : a->B;
        const uint8_t *d = frames + (size_t)f0 * a->packStride;
        if (!inPlace) {
            uint8_t *h = nullptr;
            rc = stage_area(s, &h);
            if (rc) return rc;
            memcpy(h, d, (size_t)n * a->packStride);
            d = h;
        }
        rc = async_submit(s, d, nullptr, nkpts + f0, n);
        if (rc) return rc;
    }
    return PS_OK;
}

int ps_vo_stream_pop_many(PsVoStream *s, int wait, PsHostPairResults *out)
{
    if (!s) return PS_ERR_BAD_ARG;
    int rc = bind(s->ctx);
    if (rc) return rc;
    PsVoAsync *a = s->async;
    if (!a || !out) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_pop_many: bad argument / stream not configured");
    release_view(a);
    memset(out, 0, sizeof *out);
    out->maxKpts = s->cap;
    if (a->inFlight == 0) return PS_OK;
    AsyncLane &l = a->lane[(size_t)a->head];
    if (wait) {
        PSA_HIP(hipEventSynchronize(l.evDone));
    } else {
        hipError_t q = hipEventQuery(l.evDone);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            return PS_OK;
        }
        if (q != hipSuccess) return async_fail(s, PS_ERR_HIP, "hipEventQuery", q);
    }
    uint8_t *blk = l.hres;
    if (a->mini) {
        // a mini chunk's results are a few tens of KB: copied out, so that the place keeps its pinned block (its graph writes
        // there) and is free at once
        const size_t p = (size_t)l.pairs, cap = (size_t)s->cap;
        blk = a->viewBuf.data();
        if (a->resultMode != PS_RESULTS_POSES) memcpy(blk, l.hres, p * cap * sizeof(PsDMatch));
        if (a->resultMode == PS_RESULTS_FULL) memcpy(blk + a->offMask, l.hres + a->offMask, p * cap);
        memcpy(blk + a->offPose, l.hres + a->offPose, p * 64);
        memcpy(blk + a->offStats, l.hres + a->offStats, p * sizeof(PsRansacStats));
        memcpy(blk + a->offNum, l.hres + a->offNum, p * sizeof(int32_t));
    } else {
        // the lane's block becomes the held one, the spare one the lane's: the lane is free now
        a->heldHres = l.hres;
        a->heldHresDev = l.hresDev;
        l.hres = a->spareHres;
        l.hresDev = a->spareHresDev;
        a->spareHres = a->spareHresDev = nullptr;
    }
    out->matches = a->resultMode == PS_RESULTS_POSES ? nullptr : (const PsDMatch *)blk;
    out->inlierMask = a->resultMode == PS_RESULTS_FULL ? blk + a->offMask : nullptr;
    out->resultMode = a->resultMode;
    out->pose = (const float *)(blk + a->offPose);
    out->stats = (const PsRansacStats *)(blk + a->offStats);
    out->numMatches = (const int32_t *)(blk + a->offNum);
    out->firstPair = l.firstPair;
    out->count = l.pairs;
    out->epoch = l.epoch;
    l.state = 0;
    a->head = (a->head + 1) % (int)a->lane.size();
    a->inFlight--;
    a->view = *out;
    a->haveView = true;
    return PS_OK;
}

int ps_vo_stream_pop(PsVoStream *s, int wait, PsDMatch *matches, int *nmatches, uint8_t *inlierMask, float *pose,
                     PsRansacStats *stats)
{
    if (!s) return PS_ERR_BAD_ARG;
    PsVoAsync *a = s->async;
    if (!a || !nmatches || !pose) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_pop: bad argument / stream not configured");
    *nmatches = -1;
    if (!a->haveView || a->cursor >= a->view.count) {
        PsHostPairResults v;
        // (a frame that waits in a partly filled chunk is not submitted by a pop: ps_vo_stream_flush does that)
        int rc = ps_vo_stream_pop_many(s, wait, &v);
        if (rc) return rc;
        if (v.count == 0) return PS_OK;
    }
    const PsHostPairResults &v = a->view;
    const size_t i = (size_t)a->cursor, cap = (size_t)s->cap;
    int nm = v.numMatches[i];
    if (a->resultMode == PS_RESULTS_INLIERS) nm = v.stats[i].numInliers; // the inlier matches only, as Matcher::match returns them
    if (a->resultMode == PS_RESULTS_POSES) nm = 0;
    // (checked before the cursor moves: a call refused for its arguments loses no pair)
    if (nm > 0 && (!matches || !inlierMask)) return async_fail(s, PS_ERR_BAD_ARG, "ps_vo_stream_pop: null output");
    a->cursor++;
    memcpy(pose, v.pose + i * 16, 16 * sizeof(float));
    if (stats) *stats = v.stats[i];
    if (nm > 0) {
        memcpy(matches, v.matches + i * cap, (size_t)nm * sizeof(PsDMatch));
        if (a->resultMode == PS_RESULTS_FULL)
            memcpy(inlierMask, v.inlierMask + i * cap, (size_t)nm);
        else
            memset(inlierMask, 1, (size_t)nm);
    }
    *nmatches = nm;
    return PS_OK;
}

int ps_vo_stream_pending(const PsVoStream *s)
{
    if (!s || !s->async) return PS_ERR_BAD_ARG;
    const PsVoAsync *a = s->async;
    int n = 0;
    for (const AsyncLane &l : a->lane)
        if (l.state == 1) n += l.pairs;
    if (a->haveView) n += a->view.count - a->cursor;
    if (a->staged > 0) n += a->staged - (a->prevPos >= 0 ? 0 : 1);
    return n;
}

long long ps_vo_stream_graph_launches(const PsVoStream *s)
{
    if (!s) return PS_ERR_BAD_ARG;
    return s->graphLaunches + (s->async ? s->async->graphLaunches : 0);
}

void *ps_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void ps_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

} // extern "C"
