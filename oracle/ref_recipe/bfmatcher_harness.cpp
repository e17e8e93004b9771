// bfmatcher_harness.cpp -- cv::BFMatcher(NORM_HAMMING, crossCheck = true).match(prev, cur) exactly as
// MatcherOpenCV builds and calls it (src/Matcher/matcherOpenCV.cpp:100-105,198-206) against a REAL OpenCV 3.x.
// Never built in the development image (no OpenCV there).
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <cstdint>
#include <cstdio>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = std::fopen(argv[1], "rb"), *o = std::fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int32_t cases;
    if (std::fread(&cases, 4, 1, f) != 1) return 2;
    std::fwrite(&cases, 4, 1, o);
    for (int c = 0; c < cases; ++c) {
        int32_t n;
        if (std::fread(&n, 4, 1, f) != 1) return 2;
        cv::Mat prev(n, 32, CV_8U), cur(n, 32, CV_8U);
        std::vector<float> p0((size_t)n * 3), p1((size_t)n * 3);
        if (std::fread(prev.data, 1, (size_t)n * 32, f) != (size_t)n * 32 || std::fread(cur.data, 1, (size_t)n * 32, f) != (size_t)n * 32 ||
            std::fread(p0.data(), 4, p0.size(), f) != p0.size() || std::fread(p1.data(), 4, p1.size(), f) != p1.size())
            return 2;
        cv::BFMatcher matcher(cv::NORM_HAMMING, true);
        std::vector<cv::DMatch> matches;
        matcher.match(prev, cur, matches); // query = previous frame, train = current (matcher.cpp:470-471)
        int32_t m = (int32_t)matches.size();
        std::fwrite(&m, 4, 1, o);
        for (const cv::DMatch &d : matches) {
            int32_t v[3] = {d.queryIdx, d.trainIdx, d.imgIdx};
            std::fwrite(v, 4, 3, o);
            std::fwrite(&d.distance, 4, 1, o);
        }
    }
    std::fclose(f);
    std::fclose(o);
    return 0;
}
