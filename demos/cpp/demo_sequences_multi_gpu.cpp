// demo_sequences_multi_gpu.cpp -- BASELINE configs[3] for a C++ host: a batch of independent TUM-style sequences sharded
// one per GPU of the node, every GPU running Matcher -> RANSAC/USAC -> refit on its own sequence (ps_vo_pairs_device), the
// per-pair records (pose + counts, 72 bytes) gathered on rank 0 over RCCL (include/putslam_shard.h), where the reference's
// only sequential step composes every sequence's trajectory: VO_k = VO_{k-1} * increment_k with the 0.1 m gate
// (reference src/PUTSLAM/PUTSLAM.cpp:735-740, TUM line format :1006-1016).  No torch, no Python: ONE process drives all GPUs
// (ncclCommInitAll); `--rank r --world n --id-file f` runs the same program as one process per GPU instead (ncclCommInitRank,
// the id travels through the file rank 0 writes).
//
//   demo_sequences_multi_gpu [--gpus N] [--frames 500] [--kpts 2000] [--hyp 4096] [--estimator fixed|ransac|usac]
//                            [--error-version 1] [--seed 45232] [--steps 5] [--sequence-prefix P] [--dump records.bin]
//                            [--traj-prefix T] [--repeats 1] [--warm-seconds 0] [--blocking]
//   --blocking: rounds 1 - 5's host loop (one context per GPU, ps_shard_gather_records every step) instead of the default --
//               a PsBatchQueue per GPU (two chains that are never joined) and ps_shard_gather_records_async, the records of
//               step n read after step n + 1 has been submitted.
//   --sequence-prefix P: rank r reads its frames from P<r>.bin (int32 frames, int32 cap, int32 nkpts[frames],
//                        uint8 desc[frames][cap][32], float pts[frames][cap][3]) instead of generating them.
//   --dump: rank 0 writes the gathered records, float32 [world][pairs][18], after the last step.
// Exit code 0 when every rank's records arrived and (synthetic frames) every accepted increment is within 5 mm / 5e-3 of the
// ground truth.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "putslam_dropin.h"
#include "putslam_shard.h"
#include "synth_frames.h"

namespace {

struct Sequence {
    int frames = 0, cap = 0;
    std::vector<int32_t> nk;
    std::vector<uint8_t> desc;
    std::vector<float> pts;
    std::vector<float> gt; // [frames - 1][12]: rows of the 3 x 4 ground-truth increment (synthetic frames only)
};

bool load_sequence(const std::string &path, Sequence &s)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    int32_t hdr[2];
    bool ok = std::fread(hdr, 4, 2, f) == 2 && hdr[0] >= 1 && hdr[1] >= 1 && hdr[1] <= PS_MAX_KPTS;
    if (ok) {
        s.frames = hdr[0];
        s.cap = hdr[1];
        s.nk.resize((size_t)s.frames);
        s.desc.resize((size_t)s.frames * s.cap * 32);
        s.pts.resize((size_t)s.frames * s.cap * 3);
        ok = std::fread(s.nk.data(), 4, s.nk.size(), f) == s.nk.size() && std::fread(s.desc.data(), 1, s.desc.size(), f) == s.desc.size() &&
             std::fread(s.pts.data(), 4, s.pts.size(), f) == s.pts.size();
    }
    std::fclose(f);
    return ok;
}

void make_sequence(int frames, int kpts, uint64_t seed, Sequence &s)
{
    s.frames = frames;
    s.cap = kpts;
    s.nk.assign((size_t)frames, kpts);
    s.desc.resize((size_t)frames * kpts * 32);
    s.pts.resize((size_t)frames * kpts * 3);
    s.gt.resize((size_t)(frames > 1 ? frames - 1 : 0) * 12);
    synth::World w(kpts, seed);
    for (int k = 0; k < frames; ++k) {
        w.observe(k, &s.desc[(size_t)k * kpts * 32], &s.pts[(size_t)k * kpts * 3]);
        if (k > 0) synth::increment(synth::camera_pose(k - 1), synth::camera_pose(k), &s.gt[(size_t)(k - 1) * 12]);
    }
}

struct DeviceSide { // one member's resident frames and outputs
    int device = 0;
    uint8_t *desc = nullptr;
    float *pts = nullptr;
    int32_t *nk = nullptr, *pairs = nullptr;
    static constexpr int kBlocks = 4; // consecutive steps run side by side on the member's four chains: an output block each, used in turn
    PsPairResults out[kBlocks] = {};
    void *block[kBlocks] = {};
};

#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                  \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)

int upload(const Sequence &s, int P, DeviceSide &d)
{
    HIPCHK(hipSetDevice(d.device));
    const size_t cap = (size_t)s.cap;
    HIPCHK(hipMalloc((void **)&d.desc, s.desc.size()));
    HIPCHK(hipMalloc((void **)&d.pts, s.pts.size() * 4));
    HIPCHK(hipMalloc((void **)&d.nk, s.nk.size() * 4));
    HIPCHK(hipMalloc((void **)&d.pairs, (size_t)(P > 0 ? P : 1) * 8));
    std::vector<int32_t> pairs((size_t)P * 2);
    for (int p = 0; p < P; ++p) {
        pairs[2 * (size_t)p] = p;
        pairs[2 * (size_t)p + 1] = p + 1;
    }
    HIPCHK(hipMemcpy(d.desc, s.desc.data(), s.desc.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.pts, s.pts.data(), s.pts.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.nk, s.nk.data(), s.nk.size() * 4, hipMemcpyHostToDevice));
    if (P > 0) HIPCHK(hipMemcpy(d.pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    const size_t n = (size_t)(P > 0 ? P : 1);
    const size_t bytes = n * cap * sizeof(PsDMatch) + n * cap + n * 64 + n * sizeof(PsRansacStats) + n * 4 + 256;
    for (int k = 0; k < DeviceSide::kBlocks; ++k) {
        HIPCHK(hipMalloc(&d.block[k], bytes));
        HIPCHK(hipMemset(d.block[k], 0, bytes));
        uint8_t *b = (uint8_t *)d.block[k];
        d.out[k].matches = (PsDMatch *)b;
        b += n * cap * sizeof(PsDMatch);
        d.out[k].pose = (float *)b;
        b += n * 64;
        d.out[k].stats = (PsRansacStats *)b;
        b += n * sizeof(PsRansacStats);
        d.out[k].numMatches = (int32_t *)b;
        b += n * 4;
        d.out[k].inlierMask = b;
    }
    HIPCHK(hipDeviceSynchronize()); // (the clearing above was queued on the null stream; the member's chains are not ordered with it)
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    int gpus = 0, frames = 500, kpts = 2000, hyp = 4096, errorVersion = 1, steps = 5, rank = -1, world = 0, repeats = 1;
    bool blocking = false;
    int outstanding = 4; // gathers the host keeps outstanding before it reads the oldest (--outstanding, 1 .. PS_SHARD_GATHERS_IN_FLIGHT - 1)
    double warmSeconds = 0.0;
    uint64_t seed = 0xB0B0;
    std::string estimator = "fixed", seqPrefix, dumpPath, trajPrefix, idFile;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--kpts") kpts = std::atoi(next());
        else if (a == "--hyp") hyp = std::atoi(next());
        else if (a == "--estimator") estimator = next();
        else if (a == "--error-version") errorVersion = std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
        else if (a == "--steps") steps = std::atoi(next());
        else if (a == "--repeats") repeats = std::atoi(next());
        else if (a == "--warm-seconds") warmSeconds = std::atof(next());
        else if (a == "--blocking") blocking = true;
        else if (a == "--outstanding") outstanding = std::atoi(next());
        else if (a == "--sequence-prefix") seqPrefix = next();
        else if (a == "--dump") dumpPath = next();
        else if (a == "--traj-prefix") trajPrefix = next();
        else if (a == "--rank") rank = std::atoi(next());
        else if (a == "--world") world = std::atoi(next());
        else if (a == "--id-file") idFile = next();
        else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    int avail = 0;
    if (hipGetDeviceCount(&avail) != hipSuccess || avail < 1) {
        std::fprintf(stderr, "no HIP device (there is no CPU fallback)\n");
        return 2;
    }
    PsShardGroup *g = nullptr;
    int rc;
    if (rank >= 0) { // one process per GPU
        if (world < 1 || idFile.empty()) {
            std::fprintf(stderr, "--rank needs --world and --id-file\n");
            return 2;
        }
        uint8_t id[PS_SHARD_ID_BYTES];
        if (rank == 0) {
            if (ps_shard_unique_id(id) != PS_OK) return 2;
            FILE *f = std::fopen((idFile + ".tmp").c_str(), "wb");
            if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) return 2;
            std::fclose(f);
            std::rename((idFile + ".tmp").c_str(), idFile.c_str());
        } else {
            FILE *f = nullptr;
            for (int tries = 0; tries < 600 && !(f = std::fopen(idFile.c_str(), "rb")); ++tries)
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (!f || std::fread(id, 1, sizeof id, f) != sizeof id) return 2;
            std::fclose(f);
        }
        rc = ps_shard_group_create_rank(rank % avail, rank, world, id, &g);
    } else {
        if (gpus <= 0) gpus = avail;
        if (gpus > avail) {
            std::fprintf(stderr, "--gpus %d but only %d device(s) visible\n", gpus, avail);
            return 2;
        }
        rc = ps_shard_group_create(nullptr, gpus, &g);
    }
    if (rc != PS_OK) {
        std::fprintf(stderr, "shard group: status %d\n", rc);
        return 2;
    }
    const int W = ps_shard_world_size(g), L = ps_shard_local_count(g);
    const bool driveRoot = ps_shard_rank(g, 0) == 0;

    // ---- the run's parameter block: rank 0 is authoritative (shipped defaults, resources/putslammatcherOpenCVParameters.xml:29-37)
    std::vector<PsShardRunParams> rp((size_t)L);
    for (int i = 0; i < L; ++i) {
        std::memset(&rp[(size_t)i], 0, sizeof(PsShardRunParams)); // (ranks other than 0 start empty: the broadcast fills them)
        if (ps_shard_rank(g, i) != 0) continue;
        PsShardRunParams &r = rp[(size_t)i];
        r.params.errorVersion = errorVersion;
        r.params.inlierThresholdEuclidean = 0.04;
        r.params.inlierThresholdReprojection = 2.0;
        r.params.inlierThresholdMahalanobis = 0.0002;
        r.params.minimalInlierRatioThreshold = 0.2;
        r.params.minimalNumberOfMatches = 15;
        r.params.usedPairs = 3;
        const float K[9] = {517.3f, 0.0f, 318.6f, 0.0f, 516.5f, 255.3f, 0.0f, 0.0f, 1.0f}; // freiburg1_desk.xml:5-6,20
        std::memcpy(r.K, K, sizeof K);
        r.estimator = estimator == "ransac" ? PS_EST_RANSAC : estimator == "usac" ? PS_EST_USAC : PS_EST_FIXED;
        r.numHypotheses = hyp;
        r.seed = seed;
    }
    if (ps_shard_broadcast_params(g, rp.data(), 0) != PS_OK) {
        std::fprintf(stderr, "broadcast: %s\n", ps_shard_last_error(g));
        return 2;
    }

    // ---- every member's own sequence, resident in its GPU's HBM
    std::vector<Sequence> seq((size_t)L);
    std::vector<DeviceSide> dev((size_t)L);
    int P = -1;
    for (int i = 0; i < L; ++i) {
        const int r = ps_shard_rank(g, i);
        if (!seqPrefix.empty()) {
            if (!load_sequence(seqPrefix + std::to_string(r) + ".bin", seq[(size_t)i])) {
                std::fprintf(stderr, "cannot read %s%d.bin\n", seqPrefix.c_str(), r);
                return 2;
            }
        } else {
            make_sequence(frames, kpts, 20261003ull + 7919ull * (uint64_t)r, seq[(size_t)i]);
        }
        const int p = seq[(size_t)i].frames - 1;
        if (P >= 0 && p != P) {
            std::fprintf(stderr, "sequences of different length\n");
            return 2;
        }
        P = p;
        dev[(size_t)i].device = ps_shard_device(g, i);
        if (upload(seq[(size_t)i], P, dev[(size_t)i])) return 2;
    }
    // ---- the host loop: every step submits every member's batch (asynchronous: a PsBatchQueue of four chains per member, whole
    // batches in turn; the members on their own host threads) and starts the gather of its records; the records of step n are
    // waited for after step n + 4 has been submitted, so nothing ever drains a GPU (at most PS_SHARD_GATHERS_IN_FLIGHT gathers are
    // outstanding)
    std::vector<PsRansacConfig> cfgs((size_t)L);
    std::vector<PsFrameSet> fsets((size_t)L);
    std::vector<PsShardJob> jobs((size_t)L);
    for (int i = 0; i < L; ++i) {
        const PsShardRunParams &r = rp[(size_t)i];
        PsRansacConfig &cfg = cfgs[(size_t)i];
        cfg.estimator = r.estimator;
        cfg.numHypotheses = r.numHypotheses;
        cfg.seed = r.seed + (uint64_t)ps_shard_rank(g, i); // one sequence per GPU: bench.py's seeding
        cfg.sampleIdx = nullptr;
        PsFrameSet &fs = fsets[(size_t)i];
        fs.desc = dev[(size_t)i].desc;
        fs.pts = dev[(size_t)i].pts;
        fs.nkpts = dev[(size_t)i].nk;
        fs.numFrames = seq[(size_t)i].frames;
        fs.maxKpts = seq[(size_t)i].cap;
        fs.descFrameStride = fs.ptsFrameStride = 0; // dense frames
        PsShardJob &j = jobs[(size_t)i];
        j.params = &r.params;
        j.cfg = &cfg;
        j.K = r.K;
        j.frames = &fs;
        j.pairs = dev[(size_t)i].pairs;
        j.P = P;
        j.out = &dev[(size_t)i].out[0];
    }
    std::vector<float> records(driveRoot ? (size_t)W * P * PS_SHARD_RECORD_FLOATS : 0);
    std::vector<int64_t> inFlight; // gathers started and not read yet, oldest first (four are kept outstanding)
    auto take = [&](int64_t t) -> int { // the records of gather t: complete, on the host
        const float *rec = nullptr;
        int rc2 = ps_shard_wait(g, t, &rec);
        if (rc2 != PS_OK) {
            std::fprintf(stderr, "wait: %s\n", ps_shard_last_error(g));
            return rc2;
        }
        if (driveRoot && rec) std::memcpy(records.data(), rec, records.size() * sizeof(float)); // (the consumer reads what came back)
        return PS_OK;
    };
    long long stepNo = 0;
    std::function<int()> step = [&]() -> int {
        for (int i = 0; i < L; ++i) jobs[(size_t)i].out = &dev[(size_t)i].out[stepNo % DeviceSide::kBlocks];
        ++stepNo;
        int rc2 = ps_shard_submit_all(g, jobs.data());
        if (rc2 != PS_OK) {
            std::fprintf(stderr, "submit: %s\n", ps_shard_last_error(g));
            return rc2;
        }
        // the path's only exchange: 72 bytes per pair to rank 0, behind each chain's share of the batch
        int64_t t = -1;
        rc2 = ps_shard_gather_records_async(g, P, 0, &t);
        if (rc2 != PS_OK) {
            std::fprintf(stderr, "gather: %s\n", ps_shard_last_error(g));
            return rc2;
        }
        inFlight.push_back(t);
        // the records of the step four steps back: with four chains per member four steps run side by side, and the host must
        // not wait for the oldest of them before the next one is queued (of PS_SHARD_GATHERS_IN_FLIGHT = 8 record blocks five
        // are in use)
        while ((int)inFlight.size() > outstanding) {
            if ((rc2 = take(inFlight.front())) != PS_OK) return rc2;
            inFlight.erase(inFlight.begin());
        }
        return PS_OK;
    };
    auto drain = [&]() -> int {
        int rc2 = PS_OK;
        while (!inFlight.empty() && rc2 == PS_OK) {
            rc2 = take(inFlight.front());
            inFlight.erase(inFlight.begin());
        }
        inFlight.clear();
        return rc2;
    };
    if (step() != PS_OK || drain() != PS_OK) return 2; // warm-up (code objects, stop tables, scratch arenas)
    if (blocking) {
        // rounds 1 - 5's host loop, kept for comparison: one context per member, the blocking gather every step
        std::vector<PsPairResults> results((size_t)L);
        for (int i = 0; i < L; ++i) results[(size_t)i] = dev[(size_t)i].out[0];
        step = [&, results]() -> int {
            for (int i = 0; i < L; ++i) {
                const PsShardJob &j = jobs[(size_t)i];
                PsContext *ctx = ps_shard_context(g, i);
                int rc2 = ps_vo_pairs_device(ctx, j.params, j.cfg, j.K, j.frames, j.pairs, j.P, j.out);
                if (rc2 != PS_OK) {
                    std::fprintf(stderr, "rank %d: %s\n", ps_shard_rank(g, i), ps_last_error(ctx));
                    return rc2;
                }
            }
            int rc2 = ps_shard_gather_records(g, results.data(), nullptr, P, driveRoot ? records.data() : nullptr, 0);
            if (rc2 != PS_OK) std::fprintf(stderr, "gather: %s\n", ps_shard_last_error(g));
            return rc2;
        };
        if (step() != PS_OK) return 2;
    }
    using clk = std::chrono::steady_clock;
    if (warmSeconds > 0) { // until the chip is at its steady clock, whole steps, outside the timed regions
        const auto tw = clk::now();
        while (std::chrono::duration<double>(clk::now() - tw).count() < warmSeconds)
            for (int s2 = 0; s2 < 10; ++s2)
                if (step() != PS_OK) return 2;
        if (drain() != PS_OK) return 2;
    }
    std::vector<double> rates;
    double sec = 0;
    for (int r = 0; r < (repeats > 0 ? repeats : 1); ++r) {
        const auto t0 = clk::now();
        for (int s2 = 0; s2 < steps; ++s2)
            if (step() != PS_OK) return 2;
        if (drain() != PS_OK) return 2; // the region ends when the last step's records are on the host
        sec = std::chrono::duration<double>(clk::now() - t0).count();
        rates.push_back(steps > 0 ? (double)W * P * steps / sec : 0.0);
    }

    int bad = 0;
    if (driveRoot) {
        // rank 0: the sequential step, per sequence
        for (int r = 0; r < W; ++r) {
            putslam_hip::VOTrajectory vo;
            FILE *traj = trajPrefix.empty() ? nullptr : std::fopen((trajPrefix + std::to_string(r) + ".txt").c_str(), "w");
            if (traj) std::fprintf(traj, "%s\n", putslam_hip::VOTrajectory::freiburgLine(vo.VOPoseEstimate, 1305031102.175304).c_str());
            int accepted = 0;
            double worst = 0;
            for (int p = 0; p < P; ++p) {
                const float *rec = &records[((size_t)r * P + p) * PS_SHARD_RECORD_FLOATS];
                Eigen::Matrix4f T;
                std::memcpy(T.data(), rec, 16 * sizeof(float));
                const bool identity = T(0, 0) == 1.0f && T(1, 1) == 1.0f && T(0, 3) == 0.0f && rec[16] == 0.0f;
                if (!identity) ++accepted;
                // (ground truth is known for the sequences this process generated: all of them when it drives every rank)
                if (!identity && seqPrefix.empty() && rank < 0) {
                    const float *G = &seq[(size_t)r].gt[(size_t)p * 12];
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 4; ++b) worst = std::fmax(worst, std::fabs((double)T(a, b) - (double)G[a * 4 + b]));
                }
                vo.addIncrement(T);
                if (traj)
                    std::fprintf(traj, "%s\n", putslam_hip::VOTrajectory::freiburgLine(vo.VOPoseEstimate, 1305031102.175304 + (p + 1) / 30.0).c_str());
            }
            if (traj) std::fclose(traj);
            std::printf("sequence %d: %d of %d increments accepted, worst |d| %.2e, end position (%.3f, %.3f, %.3f)\n", r, accepted, P, worst,
                        vo.VOPoseEstimate(0, 3), vo.VOPoseEstimate(1, 3), vo.VOPoseEstimate(2, 3));
            if (seqPrefix.empty() && rank < 0 && (accepted < P || worst > 5e-3)) ++bad;
            if (records[((size_t)r * P) * PS_SHARD_RECORD_FLOATS + 17] <= 0.0f) ++bad; // the rank's block never arrived
        }
        std::sort(rates.begin(), rates.end());
        std::printf("%d GPU(s) x %d pairs, %d steps x %d regions (%s): %.3f ms per step, median %.0f frame-pairs/s in all, min %.0f, max %.0f "
                    "(records gathered over RCCL every step)\n",
                    W, P, steps, (int)rates.size(), blocking ? "one chain per GPU, blocking gather" : "batch queue per GPU, asynchronous gather",
                    1e3 * sec / (steps > 0 ? steps : 1), rates[rates.size() / 2], rates.front(), rates.back());
        if (!dumpPath.empty()) {
            FILE *f = std::fopen(dumpPath.c_str(), "wb");
            if (!f || std::fwrite(records.data(), 4, records.size(), f) != records.size()) ++bad;
            if (f) std::fclose(f);
        }
    }
    ps_shard_synchronize(g);
    for (DeviceSide &d : dev) {
        (void)hipSetDevice(d.device);
        (void)hipFree(d.desc);
        (void)hipFree(d.pts);
        (void)hipFree(d.nk);
        (void)hipFree(d.pairs);
        for (void *b : d.block) (void)hipFree(b);
    }
    ps_shard_group_destroy(g);
    return bad ? 1 : 0;
}
