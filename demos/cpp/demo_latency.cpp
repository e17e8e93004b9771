// demo_latency.cpp -- what ONE call of the path costs a C / C++ host (BASELINE configs[1]: one pair of 2000-keypoint frames in the
// reference's own regime, errorVersion 0, <= 487 iterations), measured through the C ABI with std::chrono -- no Python, no
// torch, no second stream:
//   (a) ps_vo_pairs_device for one device-resident pair + ps_context_synchronize: what a caller waits for;
//   (b) the same call queued back to back on the context's stream, one synchronisation at the end: the GPU side of the
//       chain (kernels 1 - 4 of consecutive calls run in order, the host runs ahead);
//   (c) ps_vo_stream_push: Matcher::match's call shape (src/Matcher/matcher.cpp:452-516) -- host frame in, matches / mask /
//       pose / stats out, one synchronisation inside;
//   (d) the pipelined form at one frame per chunk: ps_vo_stream_push_async per frame, the result one or more calls late (every
//       place replays its chunk from a captured hipGraph; six in flight).
// usage: demo_latency [kpts=2000] [errorVersion=0] [calls=2000] [frames per chunk of leg (d) = 1]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "putslam_hip.h"
#include "synth_frames.h"

#define HIPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
            return 2;                                                                                                  \
        }                                                                                                              \
    } while (0)
#define PSCHK(x)                                                                                                       \
    do {                                                                                                               \
        if ((x) != PS_OK) {                                                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, ps_last_error(ctx));                                                  \
            return 2;                                                                                                  \
        }                                                                                                              \
    } while (0)

static double percentile(std::vector<double> v, double q)
{
    std::sort(v.begin(), v.end());
    return v[(size_t)(q * (double)(v.size() - 1))];
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int errorVersion = argc > 2 ? std::atoi(argv[2]) : 0;
    const int calls = argc > 3 ? std::atoi(argv[3]) : 2000;
    const int chunkD = argc > 4 ? std::atoi(argv[4]) : 1; // frames per chunk of leg (d)
    const int frames = 16;
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };

    synth::World world(N, 20261004);
    std::vector<uint8_t> desc((size_t)frames * N * 32);
    std::vector<float> pts((size_t)frames * N * 3);
    for (int k = 0; k < frames; ++k) world.observe(k, &desc[(size_t)k * N * 32], &pts[(size_t)k * N * 3]);

    PsContext *ctx = nullptr;
    if (ps_context_create(0, &ctx) != PS_OK) {
        std::fprintf(stderr, "ps_context_create failed (no GPU?)\n");
        return 2;
    }
    PsRansacParams prm;
    std::memset(&prm, 0, sizeof prm);
    prm.errorVersion = errorVersion; // shipped defaults, resources/putslammatcherOpenCVParameters.xml:29-37
    prm.inlierThresholdEuclidean = 0.04;
    prm.inlierThresholdReprojection = 2.0;
    prm.inlierThresholdMahalanobis = 0.0002;
    prm.minimalInlierRatioThreshold = 0.2;
    prm.minimalNumberOfMatches = 15;
    prm.usedPairs = 3;
    PsRansacConfig cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.estimator = PS_EST_RANSAC;
    cfg.numHypotheses = 487;
    cfg.seed = 3;
    const float K[9] = {517.3f, 0.0f, 318.6f, 0.0f, 516.5f, 255.3f, 0.0f, 0.0f, 1.0f};

    // ---- (a), (b): frames 0 and 1 resident in HBM
    uint8_t *dDesc = nullptr, *dMask = nullptr;
    float *dPts = nullptr, *dPose = nullptr;
    int32_t *dNk = nullptr, *dPairs = nullptr, *dNum = nullptr;
    PsDMatch *dMatches = nullptr;
    PsRansacStats *dStats = nullptr;
    HIPCHK(hipSetDevice(0));
    HIPCHK(hipMalloc((void **)&dDesc, (size_t)2 * N * 32));
    HIPCHK(hipMalloc((void **)&dPts, (size_t)2 * N * 12));
    HIPCHK(hipMalloc((void **)&dNk, 8));
    HIPCHK(hipMalloc((void **)&dPairs, 8));
    HIPCHK(hipMalloc((void **)&dMatches, (size_t)N * sizeof(PsDMatch)));
    HIPCHK(hipMalloc((void **)&dNum, 4));
    HIPCHK(hipMalloc((void **)&dMask, (size_t)N));
    HIPCHK(hipMalloc((void **)&dPose, 64));
    HIPCHK(hipMalloc((void **)&dStats, sizeof(PsRansacStats)));
    const int32_t nk[2] = {N, N}, pair[2] = {0, 1};
    HIPCHK(hipMemcpy(dDesc, desc.data(), (size_t)2 * N * 32, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dPts, pts.data(), (size_t)2 * N * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dNk, nk, 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dPairs, pair, 8, hipMemcpyHostToDevice));
    PsFrameSet fs;
    fs.desc = dDesc;
    fs.pts = dPts;
    fs.nkpts = dNk;
    fs.numFrames = 2;
    fs.maxKpts = N;
    fs.descFrameStride = fs.ptsFrameStride = 0; // dense frames
    PsPairResults out;
    out.matches = dMatches;
    out.numMatches = dNum;
    out.inlierMask = dMask;
    out.pose = dPose;
    out.stats = dStats;
    for (int i = 0; i < 200; ++i) PSCHK(ps_vo_pairs_device(ctx, &prm, &cfg, K, &fs, dPairs, 1, &out));
    PSCHK(ps_context_synchronize(ctx));
    std::vector<double> lat, enq;
    for (int i = 0; i < calls; ++i) {
        const auto t0 = clk::now();
        PSCHK(ps_vo_pairs_device(ctx, &prm, &cfg, K, &fs, dPairs, 1, &out));
        const auto t1 = clk::now();
        PSCHK(ps_context_synchronize(ctx));
        lat.push_back(us(t0, clk::now()));
        enq.push_back(us(t0, t1));
    }
    std::vector<double> chain;
    for (int turn = 0; turn < 7; ++turn) {
        const auto t0 = clk::now();
        for (int i = 0; i < 300; ++i) PSCHK(ps_vo_pairs_device(ctx, &prm, &cfg, K, &fs, dPairs, 1, &out));
        PSCHK(ps_context_synchronize(ctx));
        chain.push_back(us(t0, clk::now()) / 300);
    }
    PsRansacStats st;
    HIPCHK(hipMemcpy(&st, dStats, sizeof st, hipMemcpyDeviceToHost));
    std::printf("%d keypoints, errorVersion %d, RANSAC <= 487 (%d inliers of %d matches)\n", N, errorVersion, st.numInliers, st.numMatchesValid);
    std::printf("(a) device-resident pair, call + synchronize: median %.1f us  p10 %.1f  p90 %.1f  (host time inside the call %.1f)\n",
                percentile(lat, 0.5), percentile(lat, 0.1), percentile(lat, 0.9), percentile(enq, 0.5));
    std::printf("(b) the same call back to back, one synchronize per 300: %.1f us per pair (min %.1f)\n", percentile(chain, 0.5),
                *std::min_element(chain.begin(), chain.end()));

    // ---- (c): the per-frame streaming call
    PsVoStream *s = nullptr;
    PSCHK(ps_vo_stream_create(ctx, N, &s));
    std::vector<PsDMatch> matches((size_t)N);
    std::vector<uint8_t> mask((size_t)N);
    float pose[16];
    int nm = 0;
    std::vector<double> push;
    int accepted = 0, pushes = 0;
    for (int rep = 0; rep * frames < calls + 3 * frames; ++rep)
        for (int k = 0; k < frames; ++k) {
            cfg.seed = (uint64_t)(k + 1);
            // (the sequence is replayed forwards and backwards, so that consecutive frames are always neighbours)
            const int f = (rep & 1) ? frames - 1 - k : k;
            const auto t0 = clk::now();
            PSCHK(ps_vo_stream_push(s, &prm, &cfg, K, &desc[(size_t)f * N * 32], 32, &pts[(size_t)f * N * 3], N, matches.data(), &nm,
                                    mask.data(), pose, &st));
            const double t = us(t0, clk::now());
            if (rep >= 3) {
                push.push_back(t);
                accepted += st.accepted;
                ++pushes;
            }
        }
    std::printf("(c) ps_vo_stream_push, host frame in, results out: median %.1f us  p10 %.1f  p90 %.1f  (%d of %d increments accepted)\n",
                percentile(push, 0.5), percentile(push, 0.1), percentile(push, 0.9), accepted, pushes);
    ps_vo_stream_destroy(s);

    // ---- (d): the same frames through the PIPELINED form, one frame per chunk (the reference's call shape with the result one
    // call late): ps_vo_stream_push_async per frame, results taken as they complete.  Every place replays its chunk from a
    // captured hipGraph; six chunks are in flight.
    for (int mode = 0; mode < 2; ++mode) { // 0 = every match + mask (what (c) returns), 1 = the inlier matches (what Matcher::match returns)
        PsVoStream *ps = nullptr;
        PSCHK(ps_vo_stream_create(ctx, N, &ps));
        cfg.seed = 3;
        PSCHK(ps_vo_stream_set_result_mode(ps, mode == 0 ? PS_RESULTS_FULL : PS_RESULTS_INLIERS));
        PSCHK(ps_vo_stream_configure_async(ps, &prm, &cfg, K, chunkD, 0));
        std::vector<clk::time_point> sent;
        std::vector<double> lag;
        long long popped = 0, acc = 0;
        PsHostPairResults blk;
        auto take = [&](int wait) -> int {
            if (ps_vo_stream_pop_many(ps, wait, &blk) != PS_OK) return -1;
            if (blk.count == 0) return 0;
            const auto now = clk::now();
            for (int i = 0; i < blk.count; ++i) {
                lag.push_back(us(sent[(size_t)(blk.firstPair + i + 1)], now)); // pair k completes with frame k + 1
                acc += blk.stats[i].accepted;
            }
            popped += blk.count;
            return blk.count;
        };
        const int total = calls + 3 * frames;
        clk::time_point tStart;
        long long poppedAtStart = 0;
        for (int i = 0; i < total; ++i) {
            const int rep = i / frames, k = i % frames, f = (rep & 1) ? frames - 1 - k : k;
            if (i == 3 * frames) { // (warm: graphs captured, chip at its clock)
                tStart = clk::now();
                poppedAtStart = popped;
                lag.clear();
            }
            sent.push_back(clk::now());
            for (;;) {
                const int rc = ps_vo_stream_push_async(ps, &desc[(size_t)f * N * 32], 32, &pts[(size_t)f * N * 3], N);
                if (rc == PS_OK) break;
                if (rc != PS_ERR_BUSY || take(1) < 0) {
                    std::fprintf(stderr, "push_async: %s\n", ps_last_error(ctx));
                    return 2;
                }
            }
            while (take(0) > 0) {
            }
        }
        PSCHK(ps_vo_stream_flush(ps)); // (a partly filled last chunk)
        while (take(1) > 0) {
        }
        const double sec = std::chrono::duration<double>(clk::now() - tStart).count();
        std::printf("(d%d) ps_vo_stream_push_async, %d frame(s) per chunk, %s: %.0f frames/s, result lag median %.1f us  p90 %.1f  "
                    "(%lld chunks from graphs, %lld of %lld increments accepted)\n",
                    mode, chunkD, mode == 0 ? "every match + mask out" : "inlier matches out", (double)(popped - poppedAtStart) / sec, percentile(lag, 0.5),
                    percentile(lag, 0.9), ps_vo_stream_graph_launches(ps), acc, popped);
        if (popped != total - 1 || acc < popped - 2 * (total / frames) - 2) accepted = -1; // (every pair came back; turn-around frames aside, accepted)
        ps_vo_stream_destroy(ps);
    }
    ps_context_destroy(ctx);
    // every pushed pair but the turn-around frames (same frame twice: accepted too, identity motion) must have been accepted
    return accepted == pushes ? 0 : 1;
}
