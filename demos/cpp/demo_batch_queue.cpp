// demo_batch_queue.cpp -- what a C / C++ host that loops over batches gets from the library: BASELINE configs[2]'s 499 frame
// pairs (a 500-frame sequence resident in HBM) handed over again and again through a PsBatchQueue (include/putslam_hip.h: batches
// go to its launch chains in turn; consecutive batches write output blocks of their own) --
// the call shape of the reference's tracking loop around Matcher::match (src/PUTSLAM/PUTSLAM.cpp:677-740,
// src/Matcher/matcher.cpp:470-515), batched.  No Python, no torch, GPU_MAX_HW_QUEUES left to the library.
//
//   demo_batch_queue [--sequence f.bin] [--frames 500] [--kpts 2000] [--hyp 4096] [--estimator fixed|ransac|usac]
//                    [--error-version 1] [--seed 45232] [--steps 20] [--warmup 5] [--warm-seconds 1] [--repeats 5] [--chains 4] [--check]
//   --sequence: int32 frames, int32 cap, int32 nkpts[frames], uint8 desc[frames][cap][32], float pts[frames][cap][3]
//               (what bench.py writes for its own workload); otherwise a synthetic sequence is generated (synth_frames.h).
//   --chains 1: the same loop through ONE context (ps_vo_pairs_device), for comparison.
//   --check: the last batch's results are compared, byte for byte, with ONE ps_vo_pairs_device call on a context of its own.
// Prints one line per timed region and a last line "batch_queue: chains C, median M frame-pairs/s, ...".  Exit code 0 = ran
// (and, with --check, equal).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

namespace {

struct Results {
    void *block = nullptr;
    size_t bytes = 0;
    PsPairResults out{};
};

int alloc_results(int P, int cap, Results &r)
{
    const size_t n = (size_t)(P > 0 ? P : 1), c = (size_t)cap;
    r.bytes = n * c * sizeof(PsDMatch) + n * 64 + n * sizeof(PsRansacStats) + n * 4 + n * c + 256;
    HIPCHK(hipMalloc(&r.block, r.bytes));
    HIPCHK(hipMemset(r.block, 0, r.bytes));
    uint8_t *b = (uint8_t *)r.block;
    r.out.matches = (PsDMatch *)b;
    b += n * c * sizeof(PsDMatch);
    r.out.pose = (float *)b;
    b += n * 64;
    r.out.stats = (PsRansacStats *)b;
    b += n * sizeof(PsRansacStats);
    r.out.numMatches = (int32_t *)b;
    b += n * 4;
    r.out.inlierMask = b;
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    int frames = 500, kpts = 2000, hyp = 4096, errorVersion = 1, steps = 20, warmup = 5, repeats = 5, chains = 4, pendingWaits = 0, waitLag = 0;
    bool check = false, recordOnly = false, waitOnly = false;
    double warmSeconds = 1.0;
    uint64_t seed = 0xB0B0;
    std::string estimator = "fixed", seqPath;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--sequence") seqPath = next();
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--kpts") kpts = std::atoi(next());
        else if (a == "--hyp") hyp = std::atoi(next());
        else if (a == "--estimator") estimator = next();
        else if (a == "--error-version") errorVersion = std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
        else if (a == "--steps") steps = std::atoi(next());
        else if (a == "--warmup") warmup = std::atoi(next());
        else if (a == "--repeats") repeats = std::atoi(next());
        else if (a == "--chains") chains = std::atoi(next());
        else if (a == "--check") check = true;
        else if (a == "--pending-waits") pendingWaits = std::atoi(next());
        else if (a == "--record-only") recordOnly = true;
        else if (a == "--wait-lag") waitLag = std::atoi(next());
        else if (a == "--wait-only") waitOnly = true;
        else if (a == "--warm-seconds") warmSeconds = std::atof(next());
        else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    std::vector<int32_t> nk;
    std::vector<uint8_t> desc;
    std::vector<float> pts;
    int cap = kpts;
    if (!seqPath.empty()) {
        FILE *f = std::fopen(seqPath.c_str(), "rb");
        int32_t hdr[2];
        if (!f || std::fread(hdr, 4, 2, f) != 2 || hdr[0] < 2 || hdr[1] < 1 || hdr[1] > PS_MAX_KPTS) {
            std::fprintf(stderr, "cannot read %s\n", seqPath.c_str());
            return 2;
        }
        frames = hdr[0];
        cap = hdr[1];
        nk.resize((size_t)frames);
        desc.resize((size_t)frames * cap * 32);
        pts.resize((size_t)frames * cap * 3);
        const bool ok = std::fread(nk.data(), 4, nk.size(), f) == nk.size() && std::fread(desc.data(), 1, desc.size(), f) == desc.size() &&
                        std::fread(pts.data(), 4, pts.size(), f) == pts.size();
        std::fclose(f);
        if (!ok) {
            std::fprintf(stderr, "%s is truncated\n", seqPath.c_str());
            return 2;
        }
    } else {
        nk.assign((size_t)frames, kpts);
        desc.resize((size_t)frames * kpts * 32);
        pts.resize((size_t)frames * kpts * 3);
        synth::World w(kpts, 20261003ull);
        for (int k = 0; k < frames; ++k) w.observe(k, &desc[(size_t)k * kpts * 32], &pts[(size_t)k * kpts * 3]);
    }
    const int P = frames - 1;

    PsContext *ctx = nullptr;
    if (ps_context_create(0, &ctx) != PS_OK) {
        std::fprintf(stderr, "ps_context_create failed (no GPU? there is no CPU fallback)\n");
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
    cfg.estimator = estimator == "ransac" ? PS_EST_RANSAC : estimator == "usac" ? PS_EST_USAC : PS_EST_FIXED;
    cfg.numHypotheses = hyp;
    cfg.seed = seed;
    const float K[9] = {517.3f, 0.0f, 318.6f, 0.0f, 516.5f, 255.3f, 0.0f, 0.0f, 1.0f}; // freiburg1_desk.xml:5-6,20

    uint8_t *dDesc = nullptr;
    float *dPts = nullptr;
    int32_t *dNk = nullptr, *dPairs = nullptr;
    HIPCHK(hipMalloc((void **)&dDesc, desc.size()));
    HIPCHK(hipMalloc((void **)&dPts, pts.size() * 4));
    HIPCHK(hipMalloc((void **)&dNk, nk.size() * 4));
    HIPCHK(hipMalloc((void **)&dPairs, (size_t)P * 8));
    std::vector<int32_t> pairs((size_t)P * 2);
    for (int p = 0; p < P; ++p) {
        pairs[2 * (size_t)p] = p;
        pairs[2 * (size_t)p + 1] = p + 1;
    }
    HIPCHK(hipMemcpy(dDesc, desc.data(), desc.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dPts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dNk, nk.data(), nk.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dPairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    PsFrameSet fs;
    fs.desc = dDesc;
    fs.pts = dPts;
    fs.nkpts = dNk;
    fs.numFrames = frames;
    fs.maxKpts = cap;
    fs.descFrameStride = fs.ptsFrameStride = 0; // dense frames
    // consecutive batches run side by side on the queue's chains: an output block per chain, used in turn
    const int blocks = chains >= 2 ? chains : 1;
    std::vector<Results> outs((size_t)blocks);
    for (Results &r : outs)
        if (alloc_results(P, cap, r)) return 2;
    long long submitted = 0;
    HIPCHK(hipDeviceSynchronize()); // (the blocks' clearing was queued on the null stream; the queue's chains are not ordered with it)

    PsBatchQueue *q = nullptr;
    if (chains >= 2) PSCHK(ps_batch_queue_create(ctx, chains, &q));
    int64_t last = -1;
    auto step = [&]() -> int {
        Results &r = outs[(size_t)(submitted++ % blocks)];
        if (q) return ps_batch_queue_submit(q, &prm, &cfg, K, &fs, dPairs, P, &r.out, &last);
        return ps_vo_pairs_device(ctx, &prm, &cfg, K, &fs, dPairs, P, &r.out);
    };
    auto fence = [&]() -> int { return q ? ps_batch_queue_synchronize(q) : ps_context_synchronize(ctx); };
    using clk = std::chrono::steady_clock;
    for (int i = 0; i < warmup; ++i) PSCHK(step());
    PSCHK(fence());
    { // ... and until the chip has been busy for a second (clock ramp), whole steps, outside the timed regions
        const auto t0 = clk::now();
        while (std::chrono::duration<double>(clk::now() - t0).count() < warmSeconds) {
            for (int i = 0; i < 20; ++i) PSCHK(step());
            PSCHK(fence());
        }
    }
    std::vector<hipStream_t> aux((size_t)(pendingWaits > 0 ? (pendingWaits < 8 ? pendingWaits : 8) : 0));
    std::vector<hipEvent_t> gates((size_t)(pendingWaits > 0 ? pendingWaits * steps : 0));
    void *auxWord = nullptr;
    for (hipStream_t &a : aux) HIPCHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    for (hipEvent_t &e : gates) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (pendingWaits > 0) HIPCHK(hipMalloc(&auxWord, 256));
    std::vector<double> rates;
    double inSubmit = 0; // host seconds inside the submit calls of the timed regions
    for (int r = 0; r < repeats; ++r) {
        const auto t0 = clk::now();
        for (int i = 0; i < steps; ++i) {
            const auto s0 = clk::now();
            PSCHK(step());
            inSubmit += std::chrono::duration<double>(clk::now() - s0).count();
            // (diagnostic, --pending-waits K: K other streams each wait, device-side, for an event behind this step on its chain --
            // cross-queue waits that stay pending while the chains run: what the pipelined stream's places and the shard layer's
            // communication stream queue up, profiles/r06v/pending_waits.txt)
            if (q && pendingWaits > 0) {
                for (int k = 0; k < pendingWaits; ++k) {
                    const int c = (int)((submitted - 1) % chains);
                    hipEvent_t &ev = gates[(size_t)(i * pendingWaits + k) % gates.size()];
                    HIPCHK(hipEventRecord(ev, (hipStream_t)ps_context_stream(ps_batch_queue_context(q, c))));
                    if (recordOnly) continue; // (--record-only: the K event records alone)
                    if (i < waitLag) continue; // (--wait-lag L: the wait is queued L steps after the event it waits for was recorded)
                    hipEvent_t &evw = gates[(size_t)((i - waitLag) * pendingWaits + k) % gates.size()];
                    HIPCHK(hipStreamWaitEvent(aux[(size_t)k % aux.size()], evw, 0));
                    if (!waitOnly) HIPCHK(hipMemsetAsync(auxWord, 0, 4, aux[(size_t)k % aux.size()])); // (--wait-only: nothing behind the wait)
                }
            }
        }
        PSCHK(fence());
        for (hipStream_t a : aux) HIPCHK(hipStreamSynchronize(a));
        const double sec = std::chrono::duration<double>(clk::now() - t0).count();
        rates.push_back((double)P * steps / sec);
        std::printf("region %d: %d steps of %d pairs, %.3f ms per step, %.0f frame-pairs/s\n", r, steps, P, 1e3 * sec / steps, rates.back());
    }
    // a ticket waited for by the host: the results of that batch are complete (the call a host makes before it reads them)
    if (q) PSCHK(ps_batch_queue_wait(q, last));
    Results &res = outs[(size_t)((submitted - 1) % blocks)]; // the last batch's block
    std::vector<PsRansacStats> st((size_t)P);
    HIPCHK(hipMemcpy(st.data(), res.out.stats, st.size() * sizeof(PsRansacStats), hipMemcpyDeviceToHost));
    long long accepted = 0, inliers = 0;
    for (const PsRansacStats &s : st) {
        accepted += s.accepted;
        inliers += s.numInliers;
    }
    int bad = 0;
    if (check) {
        PsContext *solo = nullptr;
        if (ps_context_create(0, &solo) != PS_OK) return 2;
        Results ref;
        if (alloc_results(P, cap, ref)) return 2;
        if (ps_vo_pairs_device(solo, &prm, &cfg, K, &fs, dPairs, P, &ref.out) != PS_OK || ps_context_synchronize(solo) != PS_OK) {
            std::fprintf(stderr, "reference call: %s\n", ps_last_error(solo));
            return 2;
        }
        std::vector<uint8_t> a(res.bytes), b(ref.bytes);
        HIPCHK(hipMemcpy(a.data(), res.block, res.bytes, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b.data(), ref.block, ref.bytes, hipMemcpyDeviceToHost));
        // (matches beyond a pair's count and masks beyond it are unspecified: compare what is defined)
        const size_t c = (size_t)cap;
        const PsDMatch *ma = (const PsDMatch *)a.data(), *mb = (const PsDMatch *)b.data();
        const size_t offPose = (size_t)P * c * sizeof(PsDMatch), offStats = offPose + (size_t)P * 64, offNum = offStats + (size_t)P * sizeof(PsRansacStats),
                     offMask = offNum + (size_t)P * 4;
        if (std::memcmp(a.data() + offPose, b.data() + offPose, (size_t)P * 64) != 0) ++bad;
        if (std::memcmp(a.data() + offNum, b.data() + offNum, (size_t)P * 4) != 0) ++bad;
        const int32_t *num = (const int32_t *)(b.data() + offNum);
        for (int p = 0; p < P && !bad; ++p) {
            if (std::memcmp(a.data() + offStats + (size_t)p * sizeof(PsRansacStats), b.data() + offStats + (size_t)p * sizeof(PsRansacStats), 40) != 0) ++bad;
            if (std::memcmp(ma + (size_t)p * c, mb + (size_t)p * c, (size_t)num[p] * sizeof(PsDMatch)) != 0) ++bad;
            if (std::memcmp(a.data() + offMask + (size_t)p * c, b.data() + offMask + (size_t)p * c, (size_t)num[p]) != 0) ++bad;
        }
        std::printf("check against one ps_vo_pairs_device call: %s\n", bad ? "DIFFERENT" : "equal");
        (void)hipFree(ref.block);
        ps_context_destroy(solo);
    }
    std::sort(rates.begin(), rates.end());
    const char *hq = std::getenv("GPU_MAX_HW_QUEUES");
    std::printf("batch_queue: chains %d, median %.0f frame-pairs/s, min %.0f, max %.0f, %d pairs x %d steps x %d regions, accepted %lld, mean inliers %.1f, "
                "hw_queues_seen %d (GPU_MAX_HW_QUEUES=%s), host %.1f us per submit\n",
                q ? ps_batch_queue_chains(q) : 1, rates[rates.size() / 2], rates.front(), rates.back(), P, steps, repeats, accepted,
                (double)inliers / (double)(P > 0 ? P : 1), ps_context_get_option(ctx, "hw_queues_seen"), hq ? hq : "(unset)",
                1e6 * inSubmit / ((double)steps * repeats));
    if (q) ps_batch_queue_destroy(q);
    (void)hipFree(dDesc);
    (void)hipFree(dPts);
    (void)hipFree(dNk);
    (void)hipFree(dPairs);
    for (Results &r : outs) (void)hipFree(r.block);
    ps_context_destroy(ctx);
    return bad ? 1 : 0;
}
