// ransac_harness.cpp -- the reference's own RANSAC::estimateTransformation (src/TransformEst/RANSAC.cpp:50-174) on
// recorded frame pairs, built INSIDE a configured PUTSLAM build tree (CMakeLists.snippet.txt): RANSAC.cpp links g2o.
// For every case it writes the pose, the inlier list and the three match indices every iteration really sampled
// (replayed from the recorded rand() draws with the reject-duplicates rule of RANSAC.cpp:180-205), so the oracle can be
// run with exactly that sample stream.  Never built in the development image.
#include "TransformEst/RANSAC.h"

#include <opencv2/features2d/features2d.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

extern std::vector<int> g_rand_log;
extern "C" void ref_rand_reset(uint64_t seed);

static bool depthOk(const Eigen::Vector3f &p) // RANSAC.cpp:65-74
{
    return !(p.hasNaN() || p.z() < 0.1 || p.z() > 6.0);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = std::fopen(argv[1], "rb"), *o = std::fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int32_t cases;
    if (std::fread(&cases, 4, 1, f) != 1) return 2;
    std::fwrite(&cases, 4, 1, o);
    cv::Mat K = (cv::Mat_<float>(3, 3) << 517.3f, 0, 318.6f, 0, 516.5f, 255.3f, 0, 0, 1);
    for (int c = 0; c < cases; ++c) {
        int32_t n;
        if (std::fread(&n, 4, 1, f) != 1) return 2;
        cv::Mat prev(n, 32, CV_8U), cur(n, 32, CV_8U);
        std::vector<Eigen::Vector3f> p0((size_t)n), p1((size_t)n);
        if (std::fread(prev.data, 1, (size_t)n * 32, f) != (size_t)n * 32 || std::fread(cur.data, 1, (size_t)n * 32, f) != (size_t)n * 32 ||
            std::fread(p0.data(), 12, (size_t)n, f) != (size_t)n || std::fread(p1.data(), 12, (size_t)n, f) != (size_t)n)
            return 2;
        cv::BFMatcher matcher(cv::NORM_HAMMING, true);
        std::vector<cv::DMatch> matches;
        matcher.match(prev, cur, matches);
        int M = 0;
        for (const cv::DMatch &m : matches) M += (depthOk(p0[(size_t)m.queryIdx]) && depthOk(p1[(size_t)m.trainIdx])) ? 1 : 0;
        for (int mode = 0; mode < 2; ++mode) { // EUCLIDEAN_ERROR, REPROJECTION_ERROR with the shipped parameters
            RANSAC::parameters prm;
            prm.verbose = 0;
            prm.errorVersion = mode;
            prm.errorVersionVO = prm.errorVersionMap = mode;
            prm.inlierThresholdEuclidean = 0.04;
            prm.inlierThresholdReprojection = 2.0;
            prm.inlierThresholdMahalanobis = 0.0002;
            prm.minimalInlierRatioThreshold = 0.2;
            prm.minimalNumberOfMatches = 15;
            prm.usedPairs = 3;
            ref_rand_reset(1000 + 10 * (uint64_t)c + (uint64_t)mode);
            RANSAC ransac(prm, K);
            std::vector<cv::DMatch> inliers;
            Eigen::Matrix4f T = ransac.estimateTransformation(p0, p1, matches, inliers);
            // replay of getRandomMatches on the recorded draws: three distinct indices per iteration
            std::vector<int32_t> used;
            size_t pos = 0;
            while (M >= 3 && pos < g_rand_log.size()) {
                int idx[3], cnt = 0;
                while (cnt < 3 && pos < g_rand_log.size()) {
                    int v = g_rand_log[pos++] % M;
                    bool rep = false;
                    for (int i = 0; i < cnt; ++i) rep |= idx[i] == v;
                    if (!rep) idx[cnt++] = v;
                }
                if (cnt == 3) used.insert(used.end(), idx, idx + 3);
            }
            int32_t hdr[5] = {n, mode, (int32_t)matches.size(), (int32_t)inliers.size(), (int32_t)(used.size() / 3)};
            std::fwrite(hdr, 4, 5, o);
            std::fwrite(T.data(), 4, 16, o);
            for (const cv::DMatch &d : inliers) {
                int32_t v[2] = {d.queryIdx, d.trainIdx};
                std::fwrite(v, 4, 2, o);
            }
            std::fwrite(used.data(), 4, used.size(), o);
        }
    }
    std::fclose(f);
    std::fclose(o);
    return 0;
}
