// demo_matching.cpp -- C++ counterpart of the reference's demos/demoMatching.cpp for the part of its loop that
// lies on the hot path: frames enter with descriptors + back-projected 3-D points (detection is out of scope),
// go through putslam_hip::FrameMatcher (createMatcherOpenCV -> detectInitFeatures / match, src/PUTSLAM/PUTSLAM.cpp:718-740),
// the pose increments are composed with the 0.1 m gate and written as a TUM trajectory (PUTSLAM.cpp:1006-1016).
// Frames are synthetic (synth_frames.h: a static cloud seen from a camera on a smooth path, shuffled keypoint order,
// descriptor bit noise, outliers), so the estimated trajectory can be checked against the ground truth it was made from.
//
//   demo_matching [frames=100] [kpts=2000] [trajectory.txt] [--pipelined [chunk=32]]
// --pipelined: the frames go through FrameMatcher::enqueueFrame / dequeueResult (results with a lag; uploads, kernels and
//              downloads of consecutive frames overlap) instead of one synchronous runVO per frame.  Same trajectory file.
// Exit code 0 when every accepted increment is within 5 mm / 5e-3 of the ground truth.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "putslam_dropin.h"
#include "synth_frames.h"

int main(int argc, char **argv)
{
    std::vector<const char *> pos;
    bool pipelined = false, ragged = false;
    int chunk = 32;
    for (int i = 1; i < argc; ++i) {
        if (std::strcmp(argv[i], "--pipelined") == 0) {
            pipelined = true;
            if (i + 1 < argc && std::atoi(argv[i + 1]) > 0) chunk = std::atoi(argv[++i]);
        } else if (std::strcmp(argv[i], "--ragged") == 0) {
            ragged = true; // the first third of the frames carries 2/5 of the keypoints: the pipelined form has to grow
        } else {
            pos.push_back(argv[i]);
        }
    }
    const int frames = pos.size() > 0 ? std::atoi(pos[0]) : 100;
    const int N = pos.size() > 1 ? std::atoi(pos[1]) : 2000;
    const char *trajPath = pos.size() > 2 ? pos[2] : nullptr;
    synth::World world(N, 20261003);

    putslam_hip::FrameMatcher *matcher = putslam_hip::createFrameMatcher();
    matcher->setSampleSeed(42);
    if (pipelined) matcher->setPipeline(chunk, 0); // (lanes: the library's choice by chunk size)
    putslam_hip::VOTrajectory vo;
    FILE *traj = trajPath ? std::fopen(trajPath, "w") : nullptr;
    double worstT = 0, worstR = 0, seconds = 0;
    int accepted = 0, bad = 0, timed = 0, consumed = 0; // consumed: increments composed so far (= frames - 1 at the end)
    const int warmup = 3; // the first calls create the context, load the code object and build the stop tables

    auto trajLine = [&](int k) {
        if (traj) std::fprintf(traj, "%s\n", putslam_hip::VOTrajectory::freiburgLine(vo.VOPoseEstimate, 1305031102.175304 + k / 30.0).c_str());
    };
    // one increment (frame k against frame k - 1): accuracy against the ground truth, trajectory composition
    auto consume = [&](const Eigen::Matrix4f &T, const std::vector<cv::DMatch> &inliers, double ratio) {
        const int k = ++consumed;
        float G[12];
        synth::increment(synth::camera_pose(k - 1), synth::camera_pose(k), G);
        const bool identity = T(0, 0) == 1.0f && T(1, 1) == 1.0f && T(0, 3) == 0.0f && inliers.empty();
        if (!identity) {
            ++accepted;
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) worstR = std::fmax(worstR, std::fabs((double)T(r, c) - (double)G[r * 4 + c]));
                worstT = std::fmax(worstT, std::fabs((double)T(r, 3) - (double)G[r * 4 + 3]));
            }
        } else {
            ++bad;
        }
        vo.addIncrement(T);
        if (k % 20 == 0 || k == frames - 1) std::printf("frame %4d: %4zu inliers, point inlier ratio %.3f\n", k, inliers.size(), ratio);
        trajLine(k);
    };

    // one frame of the synthetic sequence, as the front end would hand it over
    auto observe = [&](int k, cv::Mat &desc, std::vector<Eigen::Vector3f> &pts) {
        desc = cv::Mat(N, 32, CV_8U);
        pts.assign((size_t)N, Eigen::Vector3f());
        world.observe(k, desc.data, reinterpret_cast<float *>(pts.data()));
        if (ragged && k < frames / 3) { // (the leading rows of the observation: a random subset of the landmarks)
            const int nk = N * 2 / 5;
            cv::Mat part(nk, 32, CV_8U);
            std::memcpy(part.data, desc.data, (size_t)nk * 32);
            desc = part;
            pts.resize((size_t)nk);
        }
    };
    // The pipelined form is timed on frames synthesised beforehand, results composed afterwards: the loop below is the matcher's calls
    // alone (synthesising a 2000-keypoint frame costs this host 0.4 ms, five to fifty times what the pipeline takes per frame).
    struct Result {
        Eigen::Matrix4f T;
        std::vector<cv::DMatch> inliers;
        double ratio;
    };
    std::vector<cv::Mat> allDesc;
    std::vector<std::vector<Eigen::Vector3f>> allPts;
    std::vector<Result> results;
    if (pipelined) {
        allDesc.resize((size_t)frames);
        allPts.resize((size_t)frames);
        for (int k = 0; k < frames; ++k) observe(k, allDesc[(size_t)k], allPts[(size_t)k]);
        results.reserve((size_t)frames);
    }
    double tEnq = 0, tWait = 0, tPoll = 0; // host seconds inside enqueueFrame / the blocking dequeue / the polling dequeues
    auto tStart = std::chrono::steady_clock::now();
    for (int k = 0; k < frames; ++k) {
        cv::Mat desc;
        std::vector<Eigen::Vector3f> pts;
        if (!pipelined) observe(k, desc, pts);
        Eigen::Matrix4f T;
        std::vector<cv::DMatch> inliers;
        double ratio = 0;
        if (pipelined) {
            if (k == 1) { // (the first frame's call created the context and built the pipeline: not part of the per-frame figures)
                tStart = std::chrono::steady_clock::now();
                tEnq = tWait = tPoll = 0;
            }
            auto now = []() { return std::chrono::steady_clock::now(); };
            auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
            auto t0 = now();
            while (!matcher->enqueueFrame(allDesc[(size_t)k], allPts[(size_t)k])) { // pipeline full: take the oldest result first
                tEnq += since(t0);
                t0 = now();
                if (matcher->dequeueResult(T, inliers, ratio, true) != 1) return 2;
                tWait += since(t0);
                results.push_back(Result{T, inliers, ratio});
                t0 = now();
            }
            tEnq += since(t0);
            t0 = now();
            while (matcher->dequeueResult(T, inliers, ratio, false) == 1) results.push_back(Result{T, inliers, ratio});
            tPoll += since(t0);
        } else if (k == 0) {
            matcher->detectInitFeatures(desc, pts);
            trajLine(0);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            ratio = matcher->runVO(desc, pts, T, inliers);
            if (k > warmup) {
                seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                ++timed;
            }
            consume(T, inliers, ratio);
        }
    }
    if (pipelined) {
        Eigen::Matrix4f T;
        std::vector<cv::DMatch> inliers;
        double ratio = 0;
        while (matcher->dequeueResult(T, inliers, ratio, true) == 1) results.push_back(Result{T, inliers, ratio});
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count(); // (from the second frame on)
        timed = frames - 1;
        trajLine(0);
        for (const Result &r : results) consume(r.T, r.inliers, r.ratio);
        std::printf("host time per frame: enqueueFrame %.1f us, blocking dequeueResult %.1f us, polling dequeueResult (+ copies) %.1f us\n",
                    1e6 * tEnq / frames, 1e6 * tWait / frames, 1e6 * tPoll / frames);
    }
    if (traj) std::fclose(traj);
    const synth::Pose last = synth::camera_pose(frames - 1);
    double pathLen = 0;
    for (int k = 1; k < frames; ++k) {
        const synth::Pose p0 = synth::camera_pose(k - 1), p1 = synth::camera_pose(k);
        pathLen += std::sqrt(std::pow(p1.t[0] - p0.t[0], 2) + std::pow(p1.t[1] - p0.t[1], 2) + std::pow(p1.t[2] - p0.t[2], 2));
    }
    const double drift = std::sqrt(std::pow(vo.VOPoseEstimate(0, 3) - last.t[0], 2) + std::pow(vo.VOPoseEstimate(1, 3) - last.t[1], 2) +
                                   std::pow(vo.VOPoseEstimate(2, 3) - last.t[2], 2));
    std::printf("%d frames x %d keypoints: %d increments accepted, %d rejected; worst |dR| %.2e, worst |dt| %.2e m; "
                "end-point drift %.4f m over %.2f m; %s %.3f ms per frame (%.0f frames/s, host frames in, pose out)\n",
                frames, N, accepted, bad, worstR, worstT, drift, pathLen,
                pipelined ? "pipelined enqueueFrame/dequeueResult" : "Matcher::runVO",
                timed ? 1e3 * seconds / timed : 0.0, timed ? timed / seconds : 0.0);
    return (consumed == frames - 1 && bad == 0 && worstR < 5e-3 && worstT < 5e-3) ? 0 : 1;
}
