// demo_matching.cpp -- C++ counterpart of the reference's demos/demoMatching.cpp for the part of its loop that
// lies on the hot path: frames enter with descriptors + back-projected 3-D points (detection is out of scope),
// go through putslam_hip::FrameMatcher (createMatcherOpenCV -> detectInitFeatures / match, src/PUTSLAM/PUTSLAM.cpp:718-740),
// the pose increments are composed with the 0.1 m gate and written as a TUM trajectory (PUTSLAM.cpp:1006-1016).
// Frames are synthetic (a static cloud seen from a camera on a smooth path, shuffled keypoint order, descriptor
// bit noise, outliers), so the estimated trajectory can be checked against the ground truth it was made from.
//
//   demo_matching [frames=100] [kpts=2000] [trajectory.txt]
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

namespace {

struct Rng { // splitmix64
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double uni(double a, double b) { return a + (b - a) * uni(); }
    double gauss() { return std::sqrt(-2.0 * std::log(uni() + 1e-300)) * std::cos(6.283185307179586 * uni()); }
};

struct Pose { // camera-to-world
    double R[3][3], t[3];
};

Pose camera_pose(int k)
{
    // smooth hand-held style path that stays near the start (the cloud must remain inside the 0.1-6 m depth window):
    // about 2 cm and 0.3 degrees per frame
    const double a = 0.12 * std::sin(0.045 * k), b = 0.15 * std::sin(0.035 * k), c = 0.10 * std::sin(0.05 * k);
    const double ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
    Pose P;
    const double Rz[3][3] = {{cc, -sc, 0}, {sc, cc, 0}, {0, 0, 1}}, Ry[3][3] = {{cb, 0, sb}, {0, 1, 0}, {-sb, 0, cb}},
                 Rx[3][3] = {{1, 0, 0}, {0, ca, -sa}, {0, sa, ca}};
    double T[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T[i][j] = 0;
            for (int m = 0; m < 3; ++m) T[i][j] += Ry[i][m] * Rx[m][j];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            P.R[i][j] = 0;
            for (int m = 0; m < 3; ++m) P.R[i][j] += Rz[i][m] * T[m][j];
        }
    P.t[0] = 0.45 * std::sin(0.04 * k);
    P.t[1] = 0.12 * std::sin(0.1 * k);
    P.t[2] = 0.35 * std::sin(0.03 * k);
    return P;
}

} // namespace

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 100;
    const int N = argc > 2 ? std::atoi(argv[2]) : 2000;
    const char *trajPath = argc > 3 ? argv[3] : nullptr;
    Rng rng(20261003);

    // static world: points in front of the start pose, one 256-bit descriptor each
    std::vector<double> world((size_t)N * 3);
    std::vector<uint8_t> wdesc((size_t)N * 32);
    for (int i = 0; i < N; ++i) {
        world[3 * i] = rng.uni(-2.5, 2.5);
        world[3 * i + 1] = rng.uni(-1.8, 1.8);
        world[3 * i + 2] = rng.uni(1.5, 5.0);
    }
    for (auto &b : wdesc) b = (uint8_t)rng.next();

    putslam_hip::FrameMatcher *matcher = putslam_hip::createFrameMatcher();
    matcher->setSampleSeed(42);
    putslam_hip::VOTrajectory vo;
    FILE *traj = trajPath ? std::fopen(trajPath, "w") : nullptr;
    Pose prevPose = camera_pose(0);
    double worstT = 0, worstR = 0, seconds = 0;
    int accepted = 0, bad = 0, timed = 0;
    const int warmup = 3; // the first calls create the context, load the code object and build the stop tables

    for (int k = 0; k < frames; ++k) {
        const Pose P = camera_pose(k);
        // observe: world -> camera k, 2 mm noise, 12 % of the keypoints replaced by clutter, random keypoint order
        std::vector<int> order((size_t)N);
        for (int i = 0; i < N; ++i) order[(size_t)i] = i;
        for (int i = N - 1; i > 0; --i) std::swap(order[(size_t)i], order[(size_t)(rng.next() % (uint64_t)(i + 1))]);
        cv::Mat desc(N, 32, CV_8U);
        std::vector<Eigen::Vector3f> pts((size_t)N);
        for (int s = 0; s < N; ++s) {
            const int i = order[(size_t)s];
            double d[3] = {world[3 * i] - P.t[0], world[3 * i + 1] - P.t[1], world[3 * i + 2] - P.t[2]}, c[3];
            for (int r = 0; r < 3; ++r) c[r] = P.R[0][r] * d[0] + P.R[1][r] * d[1] + P.R[2][r] * d[2]; // R^T d
            const bool clutter = rng.uni() < 0.12;
            uint8_t *row = desc.data + (size_t)s * 32;
            if (clutter) {
                for (int b = 0; b < 32; ++b) row[b] = (uint8_t)rng.next();
                pts[(size_t)s] = Eigen::Vector3f((float)rng.uni(-2, 2), (float)rng.uni(-1.5, 1.5), (float)rng.uni(0.5, 5.5));
            } else {
                std::memcpy(row, &wdesc[(size_t)i * 32], 32);
                for (int f = 0; f < 10; ++f) { // about 4 % of the bits flip between views
                    const unsigned bit = (unsigned)(rng.next() & 255u);
                    row[bit >> 3] ^= (uint8_t)(1u << (bit & 7u));
                }
                pts[(size_t)s] = Eigen::Vector3f((float)(c[0] + 0.002 * rng.gauss()), (float)(c[1] + 0.002 * rng.gauss()),
                                                 (float)(c[2] + 0.002 * rng.gauss()));
            }
        }
        if (k == 0) {
            matcher->detectInitFeatures(desc, pts);
        } else {
            Eigen::Matrix4f T;
            std::vector<cv::DMatch> inliers;
            const auto t0 = std::chrono::steady_clock::now();
            const double ratio = matcher->runVO(desc, pts, T, inliers);
            if (k > warmup) {
                seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                ++timed;
            }
            // ground-truth increment: current camera expressed in the previous one
            double G[3][4];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) {
                    G[r][c] = 0;
                    for (int m = 0; m < 3; ++m) G[r][c] += prevPose.R[m][r] * P.R[m][c];
                }
                G[r][3] = 0;
                for (int m = 0; m < 3; ++m) G[r][3] += prevPose.R[m][r] * (P.t[m] - prevPose.t[m]);
            }
            const bool identity = T(0, 0) == 1.0f && T(1, 1) == 1.0f && T(0, 3) == 0.0f && inliers.empty();
            if (!identity) {
                ++accepted;
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) worstR = std::fmax(worstR, std::fabs((double)T(r, c) - G[r][c]));
                    worstT = std::fmax(worstT, std::fabs((double)T(r, 3) - G[r][3]));
                }
            } else {
                ++bad;
            }
            vo.addIncrement(T);
            if (k % 20 == 0 || k == frames - 1)
                std::printf("frame %4d: %4zu inliers, point inlier ratio %.3f\n", k, inliers.size(), ratio);
        }
        if (traj) std::fprintf(traj, "%s\n", putslam_hip::VOTrajectory::freiburgLine(vo.VOPoseEstimate, 1305031102.175304 + k / 30.0).c_str());
        prevPose = P;
    }
    if (traj) std::fclose(traj);
    const Pose last = camera_pose(frames - 1);
    double pathLen = 0;
    for (int k = 1; k < frames; ++k) {
        const Pose p0 = camera_pose(k - 1), p1 = camera_pose(k);
        pathLen += std::sqrt(std::pow(p1.t[0] - p0.t[0], 2) + std::pow(p1.t[1] - p0.t[1], 2) + std::pow(p1.t[2] - p0.t[2], 2));
    }
    const double drift = std::sqrt(std::pow(vo.VOPoseEstimate(0, 3) - last.t[0], 2) + std::pow(vo.VOPoseEstimate(1, 3) - last.t[1], 2) +
                                   std::pow(vo.VOPoseEstimate(2, 3) - last.t[2], 2));
    std::printf("%d frames x %d keypoints: %d increments accepted, %d rejected; worst |dR| %.2e, worst |dt| %.2e m; "
                "end-point drift %.4f m over %.2f m; Matcher::runVO %.3f ms per frame (%.0f frames/s, host frames in, pose out)\n",
                frames, N, accepted, bad, worstR, worstT, drift,
                pathLen,
                timed ? 1e3 * seconds / timed : 0.0, timed ? timed / seconds : 0.0);
    return (bad == 0 && worstR < 5e-3 && worstT < 5e-3) ? 0 : 1;
}
