// eigen_core_harness.cpp -- the Eigen calls of src/TransformEst/RANSAC.cpp on recorded inputs, against a REAL Eigen 3.3.
// Build: see run.sh (-O2 -msse2 -DEIGEN_DONT_VECTORIZE like the reference, CMakeLists.txt:23,147).  Never built in the
// image this repository was developed in (no Eigen there).
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <cstdint>
#include <cstdio>
#include <vector>

static bool rd(FILE *f, void *p, size_t bytes) { return std::fread(p, 1, bytes, f) == bytes; }
static void wr(FILE *f, const void *p, size_t bytes) { std::fwrite(p, 1, bytes, f); }

// RANSAC::computeTransformationModel, RANSAC.cpp:207-244 (UMEYAMA branch): k x 3 MatrixXf, transposed into umeyama
static Eigen::Matrix4f umeyamaLikeReference(const float *src, const float *dst, int k)
{
    Eigen::MatrixXf features(k, 3), prevFeatures(k, 3);
    for (int i = 0; i < k; ++i)
        for (int c = 0; c < 3; ++c) {
            features(i, c) = src[3 * i + c];      // current frame = src (RANSAC.cpp:216-221)
            prevFeatures(i, c) = dst[3 * i + c];  // previous frame = dst
        }
    return Eigen::umeyama(features.transpose(), prevFeatures.transpose(), false);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = std::fopen(argv[1], "rb"), *o = std::fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int32_t hdr[3];
    if (!rd(f, hdr, sizeof hdr)) return 2;
    const int n = hdr[0], nk = hdr[1];
    std::vector<float> src((size_t)n * 9), dst((size_t)n * 9), mats((size_t)n * 9);
    if (!rd(f, src.data(), src.size() * 4) || !rd(f, dst.data(), dst.size() * 4) || !rd(f, mats.data(), mats.size() * 4)) return 2;
    wr(o, hdr, sizeof hdr);
    for (int i = 0; i < n; ++i) { // (a) 3-point umeyama, column-major 4x4; (b) its general inverse; (c) R p + t
        Eigen::Matrix4f T = umeyamaLikeReference(&src[(size_t)i * 9], &dst[(size_t)i * 9], 3);
        Eigen::Matrix4f Ti = T.inverse(); // RANSAC.cpp:337-338
        Eigen::Matrix3f R = T.block<3, 3>(0, 0);
        Eigen::Vector3f t = T.block<3, 1>(0, 3);
        Eigen::Vector3f p(src[(size_t)i * 9], src[(size_t)i * 9 + 1], src[(size_t)i * 9 + 2]);
        Eigen::Vector3f e = R * p + t; // RANSAC.cpp:266
        wr(o, T.data(), 64);
        wr(o, Ti.data(), 64);
        wr(o, e.data(), 12);
    }
    for (int i = 0; i < n; ++i) { // (d) JacobiSVD of a general 3x3 (dynamic-size matrix, as inside umeyama)
        Eigen::MatrixXf A(3, 3);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) A(r, c) = mats[(size_t)i * 9 + 3 * r + c];
        Eigen::JacobiSVD<Eigen::MatrixXf> svd(A, Eigen::ComputeFullU | Eigen::ComputeFullV);
        Eigen::Matrix3f U = svd.matrixU(), V = svd.matrixV();
        Eigen::Vector3f S = svd.singularValues();
        Eigen::Matrix<float, 3, 3, Eigen::RowMajor> Ur = U, Vr = V;
        wr(o, Ur.data(), 36);
        wr(o, S.data(), 12);
        wr(o, Vr.data(), 36);
    }
    for (int j = 0; j < nk; ++j) { // (e) k-point refits (RANSAC.cpp:153): pins Eigen's own summation order for k > 3
        int32_t k;
        if (!rd(f, &k, 4)) return 2;
        std::vector<float> s((size_t)k * 3), d((size_t)k * 3);
        if (!rd(f, s.data(), s.size() * 4) || !rd(f, d.data(), d.size() * 4)) return 2;
        Eigen::Matrix4f T = umeyamaLikeReference(s.data(), d.data(), k);
        wr(o, &k, 4);
        wr(o, T.data(), 64);
    }
    std::fclose(f);
    std::fclose(o);
    return 0;
}
