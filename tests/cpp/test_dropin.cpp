// test_dropin.cpp -- exercises the reference-shaped C++ classes (putslam_dropin.h) on the GPU.
//   test_dropin <case.bin>     case file written by tests/test_gpu_dropin.py:
//     header int32 x 8: N, M(matches), H(unused), mode, seed_lo, seed_hi, ninl, reserved
//     desc_a[N*32] desc_b[N*32] pts_a[N*3 f32] pts_b[N*3 f32] matches[M*16] pose[16 f32 col-major] mask[M]
//     kabschN int32, A[kabschN*3 f64 col-major], B[...], T[16 f64]
// Exit code 0 = every check passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "putslam_dropin.h"

static int fails = 0;
#define CHECK(c)                                                       \
    do {                                                               \
        if (!(c)) {                                                    \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c);   \
            ++fails;                                                   \
        }                                                              \
    } while (0)

template <typename T> static std::vector<T> rd(FILE *f, size_t n)
{
    std::vector<T> v(n);
    if (n && std::fread(v.data(), sizeof(T), n, f) != n) {
        std::printf("short read\n");
        std::exit(2);
    }
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::printf("usage: test_dropin case.bin\n");
        return 2;
    }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    auto hdr = rd<int32_t>(f, 8);
    const int N = hdr[0], M = hdr[1], mode = hdr[3];
    const uint64_t seed = (uint64_t)(uint32_t)hdr[4] | ((uint64_t)(uint32_t)hdr[5] << 32);
    auto da = rd<uint8_t>(f, (size_t)N * 32), db = rd<uint8_t>(f, (size_t)N * 32);
    auto pa = rd<float>(f, (size_t)N * 3), pb = rd<float>(f, (size_t)N * 3);
    auto em = rd<cv::DMatch>(f, (size_t)M);
    auto epose = rd<float>(f, 16);
    auto emask = rd<uint8_t>(f, (size_t)M);
    auto kn = rd<int32_t>(f, 1);
    auto kA = rd<double>(f, (size_t)kn[0] * 3), kB = rd<double>(f, (size_t)kn[0] * 3);
    auto kT = rd<double>(f, 16);
    std::fclose(f);

    // ---- Matcher plugin + factories (singleton ownership like matcherOpenCV.cpp:20-47) ----
    putslam_hip::FrameMatcher *matcher = putslam_hip::createFrameMatcher();
    putslam_hip::FrameMatcher *lc = putslam_hip::createLoopClosingFrameMatcher("", "");
    CHECK(matcher != nullptr && lc != nullptr && matcher != lc);
    CHECK(matcher->getName() == "OpenCV Matcher");
    cv::Mat A(N, 32, CV_8U, da.data()), B(N, 32, CV_8U, db.data());
    std::vector<cv::DMatch> matches = matcher->performMatching(A, B);
    CHECK((int)matches.size() == M);
    CHECK(M == 0 || std::memcmp(matches.data(), em.data(), (size_t)M * sizeof(cv::DMatch)) == 0);

    // ---- RANSAC with the reference signature ----
    std::vector<Eigen::Vector3f> prev((size_t)N), cur((size_t)N);
    std::memcpy((void *)prev.data(), pa.data(), (size_t)N * 12);
    std::memcpy((void *)cur.data(), pb.data(), (size_t)N * 12);
    putslam_hip::FrameMatcher::MatcherParameters mp;
    mp.RANSACParams.errorVersion = mode;
    RANSAC ransac(mp.RANSACParams, mp.cameraMatrixMat);
    ransac.setSampleSeed(seed);
    std::vector<cv::DMatch> inl;
    Eigen::Matrix4f T = ransac.estimateTransformation(prev, cur, matches, inl);
    CHECK(ransac.lastStatus() == 0);
    CHECK(std::memcmp(T.data(), epose.data(), 64) == 0);
    size_t k = 0;
    bool maskOk = true;
    for (int i = 0; i < M; ++i)
        if (emask[(size_t)i]) {
            maskOk = maskOk && k < inl.size() && inl[k].queryIdx == em[(size_t)i].queryIdx && inl[k].trainIdx == em[(size_t)i].trainIdx;
            ++k;
        }
    CHECK(maskOk && k == inl.size());
    double ratio = RANSAC::pointInlierRatio(inl, matches);
    CHECK(ratio >= 0.0 && ratio <= 1.0);

    // ---- error conventions: too few matches -> identity + cleared inliers (RANSAC.cpp:77-80) ----
    std::vector<cv::DMatch> few(matches.begin(), matches.begin() + (M < 5 ? M : 5)), inl2(3);
    Eigen::Matrix4f Ti = ransac.estimateTransformation(prev, cur, few, inl2);
    CHECK(inl2.empty());
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) CHECK(Ti(r, c) == (r == c ? 1.0f : 0.0f));

    // ---- Matcher::match data flow: prev state, errorVersionVO, state swap, pointInlierRatio ----
    matcher->setSampleSeed(seed);
    matcher->matcherParameters.RANSACParams.errorVersionVO = mode;
    matcher->detectInitFeatures(A, prev);
    Eigen::Matrix4f Tm;
    std::vector<cv::DMatch> inlm;
    double r1 = matcher->match(B, cur, Tm, inlm);
    CHECK(std::memcmp(Tm.data(), epose.data(), 64) == 0 && inlm.size() == inl.size() && r1 == ratio);
    CHECK(matcher->getNumberOfFeatures() == N);
    CHECK(matcher->getFeatures().feature3D.size() == (size_t)N && matcher->getFeatures().descriptors.rows == N);
    double r2 = matcher->match(B, cur, Tm, inlm); // cur vs cur: identity motion, all valid matches inliers
    CHECK(r2 > 0.5 && std::fabs(Tm(0, 3)) < 1e-4f && std::fabs(Tm(0, 0) - 1.0f) < 1e-5f);

    // ---- two matcher instances on two threads (main + loop closure, featuresMap.cpp:650-652) ----
    Eigen::Matrix4f Tlc;
    std::vector<cv::DMatch> inlLc;
    double rlc = 0;
    lc->matcherParameters.RANSACParams.errorVersionMap = mode;
    std::thread th([&] { rlc = lc->matchFeatureLoopClosure(A, prev, B, cur, Tlc, inlLc); });
    std::vector<cv::DMatch> m2 = matcher->performMatching(A, B);
    th.join();
    CHECK(m2.size() == matches.size() && rlc > 0.0 && inlLc.size() > 0);
    std::vector<Eigen::Vector3f> tiny(5);
    CHECK(lc->matchFeatureLoopClosure(A, tiny, B, cur, Tlc, inlLc) == 0.0); // < 10 features

    // ---- guided map matching (matchXYZ, matcher.cpp:606-798): previous frame's features as the "map" ----
    {
        std::vector<putslam_hip::FrameMatcher::MapFeatureXYZ> mapF;
        std::vector<Eigen::Vector3f> curF;
        std::vector<int> oct;
        std::vector<double> det;
        std::vector<uint8_t> curD;
        for (int i = 0; i < N; ++i) {
            if (!(prev[(size_t)i].z() > 0.1f)) continue;
            putslam_hip::FrameMatcher::MapFeatureXYZ f;
            f.id = (unsigned)i;
            for (int c = 0; c < 3; ++c) f.position[c] = prev[(size_t)i][c];
            f.descriptor = cv::Mat(1, 32, CV_8U, da.data() + (size_t)i * 32);
            f.octave = i % 4;
            f.detDist = std::sqrt(f.position[0] * f.position[0] + f.position[1] * f.position[1] + f.position[2] * f.position[2]);
            mapF.push_back(f);
            // the same physical point observed again 1 cm away, at the same scale
            curF.push_back(Eigen::Vector3f(prev[(size_t)i].x() + 0.01f, prev[(size_t)i].y(), prev[(size_t)i].z()));
            oct.push_back(i % 4);
            det.push_back(f.detDist);
            curD.insert(curD.end(), da.begin() + (size_t)i * 32, da.begin() + (size_t)(i + 1) * 32);
        }
        cv::Mat curDesc((int)curF.size(), 32, CV_8U, curD.data());
        Eigen::Matrix4f Tx;
        std::vector<cv::DMatch> inlx;
        matcher->matcherParameters.RANSACParams.errorVersionMap = 0;
        double rx = matcher->matchXYZ(mapF, curDesc, curF, oct, det, Tx, inlx, 1);
        CHECK(rx > 0.9 && inlx.size() > mapF.size() / 2);
        CHECK(std::fabs(Tx(0, 3) + 0.01f) < 1e-4f && std::fabs(Tx(1, 3)) < 1e-4f && std::fabs(Tx(0, 0) - 1.0f) < 1e-5f);
        std::vector<putslam_hip::FrameMatcher::MapFeatureXYZ> none;
        CHECK(matcher->matchXYZ(none, curDesc, curF, oct, det, Tx, inlx, 1) == -1.0);
    }

    // ---- fused resident-frame match (MatcherOpenCV) == generic performMatching -> RANSAC sequence ----
    // A subclass that switches the fused call off must return the same poses, inlier lists and ratios, frame after
    // frame, including when the frame grows past the resident store's capacity (the store is rebuilt) and when
    // detectInitFeatures restarts the sequence.
    {
        struct GenericMatcher : public putslam_hip::FrameMatcherHIP {
            GenericMatcher() { fusedMatch_ = false; }
        };
        putslam_hip::FrameMatcherHIP fusedM;
        GenericMatcher genericM;
        for (putslam_hip::FrameMatcher *m : {(putslam_hip::FrameMatcher *)&fusedM, (putslam_hip::FrameMatcher *)&genericM}) {
            m->setSampleSeed(seed ^ 0x55);
            m->matcherParameters.RANSACParams.errorVersionVO = mode;
        }
        // frames of different sizes cut from the two descriptor / point sets, then one larger than 2048 rows
        std::vector<uint8_t> bigD;
        std::vector<Eigen::Vector3f> bigP;
        for (int rep = 0; rep < 5; ++rep)
            for (int i = 0; i < N; ++i) {
                bigD.insert(bigD.end(), db.begin() + (size_t)i * 32, db.begin() + (size_t)(i + 1) * 32);
                bigD[bigD.size() - 1 - (size_t)(rep % 32)] ^= (uint8_t)(rep * 37 + 1);
                bigP.push_back(Eigen::Vector3f(cur[(size_t)i].x() + 0.001f * rep, cur[(size_t)i].y(), cur[(size_t)i].z()));
            }
        struct Frame { cv::Mat d; std::vector<Eigen::Vector3f> p; };
        std::vector<Frame> seq;
        seq.push_back({cv::Mat(N, 32, CV_8U, da.data()), prev});
        seq.push_back({cv::Mat(N, 32, CV_8U, db.data()), cur});
        seq.push_back({cv::Mat(N / 2, 32, CV_8U, da.data()), std::vector<Eigen::Vector3f>(prev.begin(), prev.begin() + N / 2)});
        seq.push_back({cv::Mat(N, 32, CV_8U, db.data()), cur});
        seq.push_back({cv::Mat((int)bigP.size(), 32, CV_8U, bigD.data()), bigP}); // 5 N rows: beyond the first store
        seq.push_back({cv::Mat(N, 32, CV_8U, da.data()), prev});
        seq.push_back({cv::Mat(N, 32, CV_8U, db.data()), cur});
        bool same = true;
        for (int pass = 0; pass < 2; ++pass) { // the second pass restarts with detectInitFeatures
            fusedM.detectInitFeatures(seq[0].d, seq[0].p);
            genericM.detectInitFeatures(seq[0].d, seq[0].p);
            for (size_t k = 1; k < seq.size(); ++k) {
                Eigen::Matrix4f Tf, Tg;
                std::vector<cv::DMatch> inf, ing;
                double rf = fusedM.match(seq[k].d, seq[k].p, Tf, inf);
                double rg = genericM.match(seq[k].d, seq[k].p, Tg, ing);
                same = same && std::memcmp(Tf.data(), Tg.data(), 64) == 0 && inf.size() == ing.size() &&
                       (rf == rg || (rf != rf && rg != rg)) &&
                       (inf.empty() || std::memcmp(inf.data(), ing.data(), inf.size() * sizeof(cv::DMatch)) == 0);
            }
        }
        CHECK(same);
    }

    // ---- USAC wrapper: same signature, no refit, best-count rule ----
    PUTSLAMEstimator::parameters up;
    up.verbose = 0;
    up.errorVersion = 0;
    up.errorVersionVO = up.errorVersionMap = 0;
    up.inlierThresholdEuclidean = 0.02; // demoUSAC.cpp:72-81
    up.inlierThresholdReprojection = 2.0;
    up.inlierThresholdMahalanobis = 0.0002;
    up.minimalInlierRatioThreshold = 0.1;
    up.usedPairs = 3;
    up.iterationCount = 0;
    RANSAC_USAC usac(up, mp.cameraMatrixMat);
    usac.setSampleSeed(seed);
    std::vector<cv::DMatch> inlu;
    Eigen::Matrix4f Tu = usac.estimateTransformation(prev, cur, matches, inlu);
    CHECK(usac.lastStatus() == 0 && inlu.size() > 10 && std::fabs(Tu(3, 3) - 1.0f) == 0.0f);

    // ---- TransformEst / KabschEst + factory (config 1: demoKabsch) ----
    putslam::TransformEst *est = putslam::createKabschEstimator();
    CHECK(est->getName() == "Kabsch Estimator");
    Eigen::MatrixXd SA(kn[0], 3), SB(kn[0], 3);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < kn[0]; ++i) {
            SA(i, c) = kA[(size_t)c * kn[0] + i];
            SB(i, c) = kB[(size_t)c * kn[0] + i];
        }
    putslam::Mat34 &tr = est->computeTransformation(SA, SB);
    double worst = 0;
#if PUTSLAM_HAVE_CV_EIGEN
    const double *td = tr.matrix().data();
#else
    const double *td = tr.data();
#endif
    for (int i = 0; i < 16; ++i) worst = std::fmax(worst, std::fabs(td[i] - kT[(size_t)i]));
    CHECK(worst < 1e-12);
    Eigen::MatrixXd E0(0, 3);
    putslam::Mat34 &t0 = est->computeTransformation(E0, E0); // empty -> identity (kabschEst.cpp:28)
#if PUTSLAM_HAVE_CV_EIGEN
    CHECK(t0.matrix()(0, 0) == 1.0 && t0.matrix()(0, 3) == 0.0);
#else
    CHECK(t0(0, 0) == 1.0 && t0(0, 3) == 0.0);
#endif

    // ---- VO trajectory driver (PUTSLAM.cpp:735-740,1006-1016) ----
    putslam_hip::VOTrajectory vo;
    vo.addIncrement(T);
    Eigen::Matrix4f big = Eigen::Matrix4f::Identity();
    big(0, 3) = 0.5f; // > 0.1 m: ignored
    vo.addIncrement(big);
    CHECK(std::memcmp(vo.VOPoseEstimate.data(), T.data(), 64) == 0);
    std::string line = putslam_hip::VOTrajectory::freiburgLine(Eigen::Matrix4f::Identity(), 1305031102.175304);
    CHECK(line == "1305031102.1753039 0 0 0 0 0 0 1");

    // ---- RGBD helpers ----
    CHECK(RGBD::roundSize(639.2, 640) == 640 && RGBD::roundSize(-1.0, 640) == 0);
    cv::Mat depth(480, 640, CV_16U);
    for (int r = 0; r < 480; ++r)
        for (int c = 0; c < 640; ++c) depth.at<uint16_t>(r, c) = 5000;
    std::vector<cv::Point2f> kp{cv::Point2f(318.6f, 255.3f), cv::Point2f(100.f, 50.f)};
    std::vector<Eigen::Vector3f> p3 = RGBD::keypoints2Dto3D(kp, depth, mp.cameraMatrixMat, 5000.0);
    CHECK(p3.size() == 2 && p3[0].x() == 0.0f && p3[0].y() == 0.0f && p3[0].z() == 1.0f);
    std::vector<cv::Point2f> back = RGBD::points3Dto2D(p3, mp.cameraMatrixMat);
    CHECK(std::fabs(back[1].x - 100.f) < 1e-3f && std::fabs(back[1].y - 50.f) < 1e-3f);

    std::printf(fails ? "test_dropin: %d FAILED\n" : "test_dropin: all checks passed\n", fails);
    return fails ? 1 : 0;
}
