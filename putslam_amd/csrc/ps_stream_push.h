// ps_stream_push.h -- the synchronous streaming form of Matcher::match (ps_vo_stream_create / _destroy / _reset / _push,
// include/putslam_hip.h).  Included by ps_capi.hip inside its extern "C" block; the pipelined form (ps_stream_async.h) builds on
// the PsVoStream defined here.
#pragma once

// ---------------------------------------------------------------------------------------------
// Streaming form of Matcher::match (reference src/Matcher/matcher.cpp:452-516): the previous frame's
// descriptors and 3-D points stay resident in HBM (the prevDescriptors / prevFeatures3D members,
// matcher.h:379-384), each push uploads only the new frame.
struct PsVoStream {
    PsContext *ctx = nullptr;
    int cap = 0;
    long long frames = 0;   // frames pushed so far
    int curSlot = 0;        // slot of the most recent frame
    int32_t nkSlot[2] = {0, 0}; // row count of the frame resident in each slot
    Buf desc, pts, meta;    // [2][cap][32], [2][cap][3], int32 {nk0, nk1, prevSlot, curSlot, seedLo, seedHi}
    // One contiguous result block on the device and its pinned host mirror, so a push needs ONE
    // device-to-host copy and ONE synchronisation: [PsRansacStats][pose 16 f32][numMatches i32 + pad]
    // [matches cap x 16 B][mask cap B]
    Buf res;
    uint8_t *hres = nullptr;   // pinned
    uint8_t *hin = nullptr;    // pinned staging of the incoming frame: [cap x 32 B][cap x 12 B][4 x i32]
    uint8_t *hresDev = nullptr, *hinDev = nullptr; // their device views (hipHostGetDevicePointer): the copy kernels' side
    size_t offPose = 0, offNum = 0, offMatches = 0, offMask = 0, resBytes = 0;
    // A push is launch-bound (three copies in, four kernels, one copy out): once the scratch
    // arena has been sized by an ordinary push with the same parameters the sequence is captured into one hipGraph
    // per frame slot and replayed with a single launch.  Everything that changes between pushes travels as data:
    // the frame (full-capacity copies from the pinned staging area), its row count and slot (meta) and the seed.
    bool graphsEnabled = true;
    bool warm = false;          // an un-captured push has run with `key`
    struct Key {
        PsRansacParams prm;
        int estimator, numHypotheses;
        float K[9];
        int options[32];               // every option of the context (kernel variants, the staged scoring's knobs), stamps
        unsigned long long arenaGen;   // PsContext::arenaGen the captured launches' pointers belong to: ANY block of the
                                       // context that is (re)allocated afterwards -- by this stream or by another call on
                                       // the same context -- invalidates the graphs
    } key{};
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    long long graphLaunches = 0;
    // PUTSLAM_HIP_PUSH_TIMING=1: host-side phases of the synchronous push, printed to stderr when the stream is destroyed
    // (staging copy | plan + tables + key | submission | wait for the GPU | results out), microseconds per push
    bool pushTiming = false;
    double pushPhase[5] = {0, 0, 0, 0, 0};
    long long pushTimed = 0;
    struct PsVoAsync *async = nullptr; // the pipelined form's state (ps_stream_async.h); null = synchronous stream
    int asyncResultMode = 0;           // PsStreamResults of the next ps_vo_stream_configure_async
    int asyncFrameLayout = 0;          // PsStreamFrames of the next ps_vo_stream_configure_async
};
static void async_release(PsVoStream *s); // (ps_stream_async.h)
static int async_reset(PsVoStream *s);

int ps_vo_stream_create(PsContext *ctx, int maxKpts, PsVoStream **out)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (!out || maxKpts < 1 || maxKpts > PS_MAX_KPTS) return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_create: bad argument");
    PsVoStream *s = new PsVoStream();
    s->ctx = ctx;
    s->cap = maxKpts;
    if (const char *v = std::getenv("PUTSLAM_HIP_PUSH_TIMING")) s->pushTiming = std::strtol(v, nullptr, 10) != 0;
    *out = s;
    const size_t cap = (size_t)maxKpts;
    s->offPose = sizeof(PsRansacStats);
    s->offNum = s->offPose + 16 * sizeof(float);
    s->offMatches = s->offNum + 16;
    s->offMask = s->offMatches + cap * sizeof(PsDMatch);
    s->resBytes = s->offMask + cap;
    PS_ENSURE(s->desc, 2 * cap * 32);
    PS_ENSURE(s->pts, 2 * cap * 12);
    PS_ENSURE(s->meta, 8 * sizeof(int32_t));
    PS_ENSURE(s->res, (s->resBytes + 3) & ~(size_t)3);
    PS_HIP(hipHostMalloc((void **)&s->hres, (s->resBytes + 3) & ~(size_t)3, hipHostMallocDefault));
    PS_HIP(hipHostMalloc((void **)&s->hin, cap * 44 + 32, hipHostMallocDefault));
    memset(s->hin, 0, cap * 44 + 32); // rows beyond a frame's count are copied by the captured graph, never read
    if (hipHostGetDevicePointer((void **)&s->hresDev, s->hres, 0) != hipSuccess || hipHostGetDevicePointer((void **)&s->hinDev, s->hin, 0) != hipSuccess) {
        (void)hipGetLastError();
        s->hresDev = s->hinDev = nullptr; // (no device view: the pushes use hipMemcpyAsync)
    }
    PS_HIP(hipMemsetAsync(s->meta.p, 0, 8 * sizeof(int32_t), ctx->stream));
    if (const char *v = std::getenv("PUTSLAM_HIP_NO_GRAPH")) s->graphsEnabled = std::atoi(v) == 0;
    return PS_OK;
}

void ps_vo_stream_destroy(PsVoStream *s)
{
    if (!s) return;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    if (s->pushTiming && s->pushTimed > 0)
        fprintf(stderr, "[putslam_hip] %lld replayed pushes, host phases in us: staging copy %.1f | plan, tables, key %.1f | submission %.1f | "
                        "wait for the GPU %.1f | results out %.1f\n", s->pushTimed, s->pushPhase[0] / s->pushTimed, s->pushPhase[1] / s->pushTimed,
                s->pushPhase[2] / s->pushTimed, s->pushPhase[3] / s->pushTimed, s->pushPhase[4] / s->pushTimed);
    async_release(s);
    for (hipGraphExec_t &g : s->gexec)
        if (g) {
            (void)hipGraphExecDestroy(g);
            g = nullptr;
        }
    Buf *all[] = {&s->desc, &s->pts, &s->meta, &s->res};
    for (Buf *b : all) release(*b);
    if (s->hres) (void)hipHostFree(s->hres);
    if (s->hin) (void)hipHostFree(s->hin);
    delete s;
}

int ps_vo_stream_reset(PsVoStream *s)
{
    if (!s) return PS_ERR_BAD_ARG;
    if (s->async) return async_reset(s);
    s->frames = 0;
    s->curSlot = 0;
    return PS_OK;
}

int ps_vo_stream_push(PsVoStream *s, const PsRansacParams *params, const PsRansacConfig *cfg, const float *K,
                      const uint8_t *desc, size_t descStep, const float *pts, int n, PsDMatch *matches, int *nmatches,
                      uint8_t *inlierMask, float *pose, PsRansacStats *stats)
{
    if (!s) return PS_ERR_BAD_ARG;
    PsContext *ctx = s->ctx;
    int rc = bind(ctx);
    if (rc) return rc;
    TimingOff toff(ctx);
    if (pose) identity16(pose);
    if (nmatches) *nmatches = 0;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->bestHypothesis = -1;
        stats->pointInlierRatio = NAN;
    }
    if (s->async) return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_push: the stream is configured for the pipelined form (push_async / push_many)");
    if (n < 0 || n > s->cap || (n > 0 && (!desc || !pts)) || descStep < PS_DESC_BYTES || !pose || !nmatches ||
        (n > 0 && (!matches || !inlierMask)) || !cfg)
        return fail(ctx, PS_ERR_BAD_ARG, "ps_vo_stream_push: bad argument");
    if (cfg->sampleIdx) return fail(ctx, PS_ERR_BAD_ARG, "explicit sample streams are not supported by the streaming call");
    const bool first = s->frames == 0;
    const int slot = first ? 0 : 1 - s->curSlot;
    const int prevSlot = s->curSlot;
    const size_t cap = (size_t)s->cap;
    using PushClock = std::chrono::steady_clock;
    PushClock::time_point tp[6];
    if (s->pushTiming) tp[0] = PushClock::now();
    // incoming frame -> pinned staging -> HBM (asynchronous; the staging area is free again after the
    // synchronisation that ends the previous push)
    uint8_t *hd = s->hin;
    float *hp = reinterpret_cast<float *>(s->hin + cap * 32);
    // meta block exactly as it lies on the device: {nk[slot 0], nk[slot 1], prevSlot, slot, seedLo, seedHi} -- ONE copy per push
    // (round 3 sent the row count, the slot pair and the seed as three copies: each is a node of the captured graph with a few
    // microseconds of its own)
    int32_t *hm = reinterpret_cast<int32_t *>(s->hin + cap * 44);
    for (int i = 0; i < n; ++i) memcpy(hd + (size_t)i * 32, desc + (size_t)i * descStep, 32);
    if (n > 0) memcpy(hp, pts, (size_t)n * 12);
    hm[slot] = n;
    hm[1 - slot] = s->nkSlot[1 - slot];
    hm[2] = prevSlot; // query = previous frame, train = current (matcher.cpp:470-471)
    hm[3] = slot;
    memcpy(&hm[4], &cfg->seed, sizeof(uint64_t));
    if (s->pushTiming) tp[1] = PushClock::now();
    // The stream's state (curSlot, frames) is committed only when the push has succeeded: after a failed push
    // (bad parameters, a HIP error) the resident frame is still the previous one and the next push matches against it.
    auto commit = [&]() {
        s->curSlot = slot;
        s->nkSlot[slot] = n;
        s->frames++;
    };
    // Frame in / results out as ONE kernel each over the mapped pinned staging blocks (ps_copy_segments) instead of three and one
    // hipMemcpyAsync: a copy of this size is a node of its own with 5 - 8 us of latency in the captured graph, the kernel reads
    // the 88 KB of a 2000-keypoint frame over the link in 4 (option "stream_copy_kernels" = 0: the copies of rounds 1 - 4).
    const bool copyKernels = ctx->streamCopyKernels != 0 && s->hinDev != nullptr && s->hresDev != nullptr;
    auto copy_in = [&](size_t rows) -> int {
        if (copyKernels) {
            CopySegs up{};
            int k = 0;
            if (rows > 0) {
                up.src[k] = s->hinDev;
                up.dst[k] = (uint8_t *)s->desc.p + (size_t)slot * cap * 32;
                up.bytes[k++] = rows * 32;
                up.src[k] = s->hinDev + cap * 32;
                up.dst[k] = (uint8_t *)s->pts.p + (size_t)slot * cap * 12;
                up.bytes[k++] = rows * 12;
            }
            up.src[k] = s->hinDev + cap * 44;
            up.dst[k] = s->meta.p;
            up.bytes[k++] = 6 * sizeof(int32_t);
            up.n = k;
            const unsigned groups = (unsigned)((rows * 32 / 16 + 255) / 256);
            hipLaunchKernelGGL(ps_copy_segments, dim3(groups < 1 ? 1 : (groups > 64 ? 64 : groups)), dim3(256), 0, ctx->stream, up);
            PS_HIP(hipGetLastError());
            return PS_OK;
        }
        if (rows > 0) {
            PS_HIP(hipMemcpyAsync((uint8_t *)s->desc.p + (size_t)slot * cap * 32, hd, rows * 32, hipMemcpyHostToDevice,
                                  ctx->stream));
            PS_HIP(hipMemcpyAsync((float *)s->pts.p + (size_t)slot * cap * 3, hp, rows * 12, hipMemcpyHostToDevice,
                                  ctx->stream));
        }
        PS_HIP(hipMemcpyAsync(s->meta.p, hm, 6 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        return PS_OK;
    };
    if (first) { // detectInitFeatures (matcher.cpp:17-64): nothing to match against yet
        rc = copy_in((size_t)n);
        if (rc) return rc;
        PS_HIP(hipStreamSynchronize(ctx->stream));
        commit();
        *nmatches = -1;
        return PS_OK;
    }
    PsFrameSet fs;
    fs.desc = (const uint8_t *)s->desc.p;
    fs.pts = (const float *)s->pts.p;
    fs.nkpts = (const int32_t *)s->meta.p;
    fs.numFrames = 2;
    fs.maxKpts = s->cap;
    fs.descFrameStride = fs.ptsFrameStride = 0;
    Plan pl;
    rc = make_plan(ctx, params, cfg, K, s->cap, s->cap, pl);
    if (rc) return rc;
    pl.ma.seedDev = reinterpret_cast<const uint64_t *>((const int32_t *)s->meta.p + 4);
    rc = prepare_score(ctx, pl, 1, s->cap);
    if (rc) return rc;
    // (outside the capture: a captured push then holds no clearing node, and a replay finds the block as its capture did)
    PS_ENSURE(ctx->keys, (size_t)s->cap * sizeof(uint32_t));
    rc = keys_clean(ctx, (size_t)s->cap * sizeof(uint32_t));
    if (rc) return rc;
    uint8_t *dres = (uint8_t *)s->res.p;
    auto enqueue = [&](size_t rows) -> int {
        int r = copy_in(rows);
        if (r) return r;
        r = run_match_stage(ctx, fs, (const int32_t *)s->meta.p + 2, 1, true, pl.pa, (PsDMatch *)(dres + s->offMatches),
                            (int32_t *)(dres + s->offNum), 0);
        if (r) return r;
        r = run_ransac_stage(ctx, pl, 1, s->cap, (const PsDMatch *)(dres + s->offMatches),
                             (const int32_t *)(dres + s->offNum), s->cap, (float *)(dres + s->offPose), dres + s->offMask,
                             (PsRansacStats *)dres, 2);
        if (r) return r;
        if (copyKernels) {
            CopySegs down{};
            down.src[0] = dres;
            down.dst[0] = s->hresDev;
            down.bytes[0] = (s->resBytes + 3) & ~(size_t)3;
            down.n = 1;
            const unsigned groups = (unsigned)((s->resBytes / 16 + 255) / 256);
            hipLaunchKernelGGL(ps_copy_segments, dim3(groups < 1 ? 1 : (groups > 32 ? 32 : groups)), dim3(256), 0, ctx->stream, down);
            PS_HIP(hipGetLastError());
            return PS_OK;
        }
        PS_HIP(hipMemcpyAsync(s->hres, dres, s->resBytes, hipMemcpyDeviceToHost, ctx->stream));
        return PS_OK;
    };
    PsVoStream::Key key;
    memset(&key, 0, sizeof key);
    // field by field: the caller's struct may carry indeterminate padding bytes, the key is compared with memcmp
    key.prm.verbose = params->verbose;
    key.prm.errorVersion = params->errorVersion;
    key.prm.errorVersionVO = params->errorVersionVO;
    key.prm.errorVersionMap = params->errorVersionMap;
    key.prm.inlierThresholdEuclidean = params->inlierThresholdEuclidean;
    key.prm.inlierThresholdReprojection = params->inlierThresholdReprojection;
    key.prm.inlierThresholdMahalanobis = params->inlierThresholdMahalanobis;
    key.prm.minimalInlierRatioThreshold = params->minimalInlierRatioThreshold;
    key.prm.minimalNumberOfMatches = params->minimalNumberOfMatches;
    key.prm.usedPairs = params->usedPairs;
    key.prm.iterationCount = params->iterationCount;
    {
        int n = psi_options_snapshot(ctx, key.options, (int)(sizeof key.options / sizeof key.options[0]) - 1);
        key.options[n++] = ctx->stampsOn;
    }
    key.estimator = cfg->estimator;
    key.numHypotheses = cfg->numHypotheses;
    if (K) memcpy(key.K, K, sizeof key.K);
    key.arenaGen = ctx->arenaGen; // (make_plan / prepare_score above may already have grown a block: then no replay)
    const bool sameKey = s->warm && memcmp(&key, &s->key, sizeof key) == 0;
    if (!sameKey) { // new parameters: the next ordinary push re-sizes scratch and tables, graphs are rebuilt after it
        for (hipGraphExec_t &g : s->gexec)
            if (g) {
                (void)hipGraphExecDestroy(g);
                g = nullptr;
            }
    }
    bool launched = false;
    if (s->pushTiming) tp[2] = PushClock::now();
    if (s->graphsEnabled && sameKey) {
        if (!s->gexec[slot]) {
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int r = enqueue(cap);
                hipError_t e2 = hipStreamEndCapture(ctx->stream, &graph);
                if (r == PS_OK && e2 == hipSuccess && graph &&
                    hipGraphInstantiate(&s->gexec[slot], graph, nullptr, nullptr, 0) != hipSuccess)
                    s->gexec[slot] = nullptr;
                if (r != PS_OK || e2 != hipSuccess) s->gexec[slot] = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (!s->gexec[slot]) {
                s->graphsEnabled = false; // capture is not available here: stay on ordinary launches
                (void)hipGetLastError();
                ctx->err.clear();
            }
        }
        if (s->gexec[slot]) {
            PS_HIP(hipGraphLaunch(s->gexec[slot], ctx->stream));
            s->graphLaunches++;
            launched = true;
        }
    }
    if (!launched) {
        rc = enqueue((size_t)n);
        if (rc) return rc;
        key.arenaGen = ctx->arenaGen; // the blocks as this ordinary push left them
        memcpy(&s->key, &key, sizeof key); // (bytewise, padding included: the key is compared with memcmp)
        s->warm = true;
    }
    if (s->pushTiming) tp[3] = PushClock::now();
    PS_HIP(hipStreamSynchronize(ctx->stream));
    if (s->pushTiming) tp[4] = PushClock::now();
    commit();
    int32_t nm = 0;
    memcpy(&nm, s->hres + s->offNum, sizeof nm);
    memcpy(pose, s->hres + s->offPose, 16 * sizeof(float));
    if (stats) memcpy(stats, s->hres, sizeof *stats);
    if (nm > 0) {
        memcpy(matches, s->hres + s->offMatches, (size_t)nm * sizeof(PsDMatch));
        memcpy(inlierMask, s->hres + s->offMask, (size_t)nm);
    }
    *nmatches = nm;
    if (s->pushTiming && launched) {
        tp[5] = PushClock::now();
        for (int i = 0; i < 5; ++i) s->pushPhase[i] += std::chrono::duration<double, std::micro>(tp[i + 1] - tp[i]).count();
        s->pushTimed++;
    }
    return PS_OK;
}

