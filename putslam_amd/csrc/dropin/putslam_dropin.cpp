// putslam_dropin.cpp -- reference-shaped C++ classes over the C ABI (see putslam_dropin.h).
// Host code only: marshals std::vector / cv::Mat / Eigen storage into plain pointers and calls
// libputslam_hip.so.  One PsContext (HIP stream + scratch arena) per calling thread, because the
// reference constructs a fresh RANSAC object per call (matcher.cpp:493) and runs a second Matcher on
// the loop-closure thread (featuresMap.cpp:650-652): contexts are never shared between threads.
#include "putslam_dropin.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <iostream>
#include <sstream>
#include <iomanip>

#include "putslam_hip.h"

static_assert(sizeof(cv::DMatch) == sizeof(PsDMatch), "cv::DMatch layout");
static_assert(sizeof(Eigen::Vector3f) == 12, "Eigen::Vector3f storage");

namespace {

struct ThreadContext {
    PsContext *ctx = nullptr;
    int status = PS_OK;
    ThreadContext()
    {
        int dev = 0;
        if (const char *e = std::getenv("PUTSLAM_HIP_DEVICE")) dev = std::atoi(e);
        status = ps_context_create(dev, &ctx);
        if (status != PS_OK)
            std::cerr << "putslam_hip: no usable HIP device (status " << status << "); the GPU path has no CPU fallback"
                      << std::endl;
    }
    ~ThreadContext() { ps_context_destroy(ctx); }
};

PsContext *threadContext(int *status)
{
    static thread_local ThreadContext tc;
    if (status) *status = tc.status;
    return tc.ctx;
}

void cameraToK(const cv::Mat &cameraMatrix, float K[9], bool &have)
{
    have = !cameraMatrix.empty() && cameraMatrix.rows >= 3 && cameraMatrix.cols >= 3;
    for (int i = 0; i < 9; ++i) K[i] = 0.0f;
    if (have)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) K[3 * r + c] = cameraMatrix.at<float>(r, c);
}

PsRansacParams toPs(const RANSAC::parameters &p)
{
    PsRansacParams q;
    std::memset(&q, 0, sizeof q); // padding bytes too: the C ABI may compare parameter blocks bytewise
    q.verbose = p.verbose;
    q.errorVersion = p.errorVersion;
    q.errorVersionVO = p.errorVersionVO;
    q.errorVersionMap = p.errorVersionMap;
    q.inlierThresholdEuclidean = p.inlierThresholdEuclidean;
    q.inlierThresholdReprojection = p.inlierThresholdReprojection;
    q.inlierThresholdMahalanobis = p.inlierThresholdMahalanobis;
    q.minimalInlierRatioThreshold = p.minimalInlierRatioThreshold;
    q.minimalNumberOfMatches = p.minimalNumberOfMatches;
    q.usedPairs = p.usedPairs;
    q.iterationCount = p.iterationCount;
    return q;
}

Eigen::Matrix4f runEstimator(int estimator, int H, uint64_t seed, const PsRansacParams &prm, const cv::Mat &cameraMatrix,
                             const std::vector<Eigen::Vector3f> &prev, const std::vector<Eigen::Vector3f> &cur,
                             const std::vector<cv::DMatch> &matches, std::vector<cv::DMatch> &inliers, bool clearOnFail,
                             int &status)
{
    Eigen::Matrix4f pose = Eigen::Matrix4f::Identity();
    PsContext *ctx = threadContext(&status);
    if (!ctx) {
        if (clearOnFail) inliers.clear();
        return pose;
    }
    float K[9];
    bool haveK;
    cameraToK(cameraMatrix, K, haveK);
    PsRansacConfig cfg;
    cfg.estimator = estimator;
    cfg.numHypotheses = H;
    cfg.seed = seed;
    cfg.sampleIdx = nullptr;
    std::vector<cv::DMatch> out(matches.size() ? matches.size() : 1);
    int n = 0;
    PsRansacStats st;
    status = ps_ransac_rigid3d(ctx, &prm, &cfg, haveK ? K : nullptr, reinterpret_cast<const float *>(prev.data()),
                               (int)prev.size(), reinterpret_cast<const float *>(cur.data()), (int)cur.size(),
                               reinterpret_cast<const PsDMatch *>(matches.data()), (int)matches.size(), pose.data(),
                               reinterpret_cast<PsDMatch *>(out.data()), &n, nullptr, &st);
    if (status != PS_OK) {
        std::cerr << "putslam_hip: " << ps_last_error(ctx) << std::endl;
        pose = Eigen::Matrix4f::Identity();
        if (clearOnFail) inliers.clear();
        return pose;
    }
    if (estimator == PS_EST_USAC && st.numMatchesValid < 8) return pose; // USAC_wrapper.cpp:120-122: inliers untouched
    out.resize((size_t)n);
    inliers.swap(out);
    if (prm.verbose > 0) {
        std::cout << "RANSAC: matches.size() = " << st.numMatchesValid << std::endl;
        std::cout << "RANSAC best model : inlierRatio = " << st.bestInlierRatio * 100.0 << "%" << std::endl;
    }
    return pose;
}

int ransacIterations(double inlierRatio, double successProbability = 0.98, int numberOfPairs = 3)
{
    // RANSAC::computeRANSACIteration, RANSAC.cpp:457-461 (saturating instead of UB)
    double v = std::log(1 - successProbability) / std::log(1 - std::pow(inlierRatio, numberOfPairs));
    if (!(v < 2147483647.0)) return 2147483647;
    return v < 0 ? 0 : (int)v;
}

} // namespace

// ---------------------------------------------------------------------------------------------
RANSAC::RANSAC(RANSAC::parameters _RANSACParameters, cv::Mat _cameraMatrix)
{
    seed_ = (uint64_t)std::time(nullptr); // the reference: srand(time(0)), RANSAC.cpp:13
    cameraMatrix = _cameraMatrix;
    RANSACParams = _RANSACParameters;
    RANSACParams.iterationCount = ransacIterations(0.20); // RANSAC.cpp:30
    if (RANSACParams.verbose > 0) {
        std::cout << "RANSACParams.verbose --> " << RANSACParams.verbose << std::endl;
        std::cout << "RANSACParams.usedPairs --> " << RANSACParams.usedPairs << std::endl;
        std::cout << "RANSACParams.errorVersion --> " << RANSACParams.errorVersion << std::endl;
        std::cout << "RANSACParams.inlierThresholdEuclidean --> " << RANSACParams.inlierThresholdEuclidean << std::endl;
        std::cout << "RANSACParams.inlierThresholdReprojection --> " << RANSACParams.inlierThresholdReprojection << std::endl;
        std::cout << "RANSACParams.minimalInlierRatioThreshold --> " << RANSACParams.minimalInlierRatioThreshold << std::endl;
    }
}

Eigen::Matrix4f RANSAC::estimateTransformation(std::vector<Eigen::Vector3f> prevFeatures, std::vector<Eigen::Vector3f> features,
                                               std::vector<cv::DMatch> matches, std::vector<cv::DMatch> &bestInlierMatches)
{
    if (RANSACParams.verbose > 0) std::cout << "RANSAC: original matches.size() = " << matches.size() << std::endl;
    if (RANSACParams.errorVersion != EUCLIDEAN_ERROR && RANSACParams.errorVersion != REPROJECTION_ERROR &&
        RANSACParams.errorVersion != EUCLIDEAN_AND_REPROJECTION_ERROR && RANSACParams.errorVersion != ADAPTIVE_ERROR)
        std::cout << "RANSAC: incorrect error version" << std::endl; // RANSAC.cpp:134-135 (once per call here, not per iteration)
    // enough samples for the longest schedule the reference can run: max(iters(0.2), iters(minRatio))
    int a = ransacIterations(0.20), b = ransacIterations(RANSACParams.minimalInlierRatioThreshold);
    int H = a > b ? a : b;
    if (H < 1) H = 1;
    if (H > PS_MAX_HYPOTHESES) H = PS_MAX_HYPOTHESES;
    PsRansacParams prm = toPs(RANSACParams);
    return runEstimator(PS_EST_RANSAC, H, seed_, prm, cameraMatrix, prevFeatures, features, matches, bestInlierMatches, true,
                        lastStatus_);
}

// ---------------------------------------------------------------------------------------------
RANSAC_USAC::RANSAC_USAC(PUTSLAMEstimator::parameters p, cv::Mat _cameraMatrix)
{
    seed_ = (uint64_t)std::time(nullptr); // USAC_wrapper.cpp:16
    cameraMatrix = _cameraMatrix;
    params_ = p;
}
RANSAC_USAC::~RANSAC_USAC() {}

Eigen::Matrix4f RANSAC_USAC::estimateTransformation(std::vector<Eigen::Vector3f> prevFeatures, std::vector<Eigen::Vector3f> features,
                                                    std::vector<cv::DMatch> matches, std::vector<cv::DMatch> &bestInlierMatches)
{
    PsRansacParams prm;
    std::memset(&prm, 0, sizeof prm);
    prm.verbose = params_.verbose;
    prm.errorVersion = params_.errorVersion;
    prm.errorVersionVO = params_.errorVersionVO;
    prm.errorVersionMap = params_.errorVersionMap;
    prm.inlierThresholdEuclidean = params_.inlierThresholdEuclidean;
    prm.inlierThresholdReprojection = params_.inlierThresholdReprojection;
    prm.inlierThresholdMahalanobis = params_.inlierThresholdMahalanobis;
    prm.minimalInlierRatioThreshold = params_.minimalInlierRatioThreshold;
    prm.minimalNumberOfMatches = 8; // USAC_wrapper.cpp:120
    prm.usedPairs = params_.usedPairs;
    return runEstimator(PS_EST_USAC, maxHyp_, seed_, prm, cameraMatrix, prevFeatures, features, matches, bestInlierMatches, false,
                        lastStatus_);
}

// ---------------------------------------------------------------------------------------------
int RGBD::roundSize(double x, int size)
{
    if (x < 0)
        x = 0;
    else if (x > size - 1)
        x = size;
    return (int)std::round(x);
}

std::vector<Eigen::Vector3f> RGBD::keypoints2Dto3D(std::vector<cv::Point2f> f2d, cv::Mat depthImage, cv::Mat cameraMatrix,
                                                   double depthImageScale, int startingID)
{
    size_t n = f2d.size() > (size_t)startingID ? f2d.size() - (size_t)startingID : 0;
    std::vector<Eigen::Vector3f> out(n);
    if (n == 0) return out;
    int status;
    PsContext *ctx = threadContext(&status);
    float K[9];
    bool haveK;
    cameraToK(cameraMatrix, K, haveK);
    if (!ctx) return out;
    status = ps_keypoints2Dto3D(ctx, reinterpret_cast<const float *>(f2d.data() + startingID), (int)n,
                                reinterpret_cast<const uint16_t *>(depthImage.data), depthImage.rows, depthImage.cols,
                                (size_t)depthImage.step, K, depthImageScale, reinterpret_cast<float *>(out.data()));
    if (status != PS_OK) std::cerr << "putslam_hip: " << ps_last_error(ctx) << std::endl;
    return out;
}

std::vector<cv::Point2f> RGBD::removeImageDistortion(std::vector<cv::Point2f> &features, cv::Mat cameraMatrix, cv::Mat distCoeffs)
{
    if (features.size() == 0) return std::vector<cv::Point2f>(); // RGBD.cpp:258-259
    std::vector<cv::Point2f> out(features.size());
    int status;
    PsContext *ctx = threadContext(&status);
    float K[9];
    bool haveK;
    cameraToK(cameraMatrix, K, haveK);
    double d[5] = {0, 0, 0, 0, 0};
    const int nd = distCoeffs.empty() ? 0 : distCoeffs.rows * distCoeffs.cols;
    for (int i = 0; i < 5 && i < nd; ++i) d[i] = (double)distCoeffs.at<float>(distCoeffs.rows == 1 ? 0 : i, distCoeffs.rows == 1 ? i : 0);
    if (!ctx) return out;
    status = ps_remove_image_distortion(ctx, reinterpret_cast<const float *>(features.data()), (int)features.size(), K, d,
                                        reinterpret_cast<float *>(out.data()));
    if (status != PS_OK) std::cerr << "putslam_hip: " << ps_last_error(ctx) << std::endl;
    return out;
}

std::vector<cv::Point2f> RGBD::points3Dto2D(std::vector<Eigen::Vector3f> f3d, cv::Mat cameraMatrix)
{
    std::vector<cv::Point2f> out(f3d.size());
    if (f3d.empty()) return out;
    int status;
    PsContext *ctx = threadContext(&status);
    float K[9];
    bool haveK;
    cameraToK(cameraMatrix, K, haveK);
    if (!ctx) return out;
    status = ps_points3Dto2D(ctx, reinterpret_cast<const float *>(f3d.data()), (int)f3d.size(), K,
                             reinterpret_cast<float *>(out.data()));
    if (status != PS_OK) std::cerr << "putslam_hip: " << ps_last_error(ctx) << std::endl;
    return out;
}

namespace putslam {

// ---------------------------------------------------------------------------------------------
KabschEst::Ptr kabsch; // "A single instance of Kabsch Estimator", kabschEst.cpp:8

KabschEst::KabschEst(void) : name("Kabsch Estimator") {}
const std::string &KabschEst::getName() const { return name; }

Mat34 &KabschEst::computeTransformation(const Eigen::MatrixXd &setA, const Eigen::MatrixXd &setB)
{
    transformation.setIdentity();
    if (setA.rows() == 0) return transformation; // kabschEst.cpp:28
    int status;
    PsContext *ctx = threadContext(&status);
    if (!ctx) return transformation;
    double T[16];
    status = ps_kabsch_f64(ctx, setA.data(), setB.data(), (int)setA.rows(), (int)setA.rows(), T);
    if (status != PS_OK) {
        std::cerr << "putslam_hip: " << ps_last_error(ctx) << std::endl;
        return transformation;
    }
#if PUTSLAM_HAVE_CV_EIGEN
    std::memcpy(transformation.matrix().data(), T, sizeof T);
#else
    std::memcpy(transformation.data(), T, sizeof T);
#endif
    return transformation;
}

TransformEst *createKabschEstimator(void)
{
    kabsch.reset(new KabschEst());
    return kabsch.get();
}

} // namespace putslam

namespace putslam_hip {

// ---------------------------------------------------------------------------------------------
FrameMatcher::MatcherParameters::MatcherParameters()
{
    // shipped defaults, resources/putslammatcherOpenCVParameters.xml:29-37
    RANSACParams.verbose = 0;
    RANSACParams.errorVersion = 0;
    RANSACParams.errorVersionVO = 0;
    RANSACParams.errorVersionMap = 0;
    RANSACParams.inlierThresholdEuclidean = 0.04;
    RANSACParams.inlierThresholdReprojection = 2.0;
    RANSACParams.inlierThresholdMahalanobis = 0.0002;
    RANSACParams.minimalInlierRatioThreshold = 0.2;
    RANSACParams.minimalNumberOfMatches = 15;
    RANSACParams.usedPairs = 3;
    RANSACParams.iterationCount = 0;
    // resources/datasetConfig/freiburg1_desk.xml:5-6 (also hard-coded at RGBD.cpp:22-23)
    cameraMatrixMat = cv::Mat(3, 3, CV_32FC1);
    const float K[9] = {517.3f, 0.0f, 318.6f, 0.0f, 516.5f, 255.3f, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < 9; ++i) cameraMatrixMat.at<float>(i / 3, i % 3) = K[i];
}

void FrameMatcher::detectInitFeatures(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D)
{
    prevDescriptors = descriptors;
    prevFeatures3D.swap(features3D);
    frameCounter = 0;
    fusedSynced_ = false;
}

// Resident-frame state of one matcher instance: its own context (stream + scratch) and the two-slot frame store.
struct FrameMatcher::Fused {
    PsContext *ctx = nullptr;
    PsVoStream *stream = nullptr;
    int cap = 0;
    PsVoStream *pipe = nullptr; // the pipelined form's stream (enqueueFrame / dequeueResult)
    int pipeCap = 0;
    std::vector<cv::DMatch> pm;
    std::vector<uint8_t> pmask;
    ~Fused()
    {
        if (pipe) ps_vo_stream_destroy(pipe);
        if (stream) ps_vo_stream_destroy(stream);
        if (ctx) ps_context_destroy(ctx);
    }
};

FrameMatcher::FrameMatcher(const std::string _name) : name(_name), frameCounter(0) {}
FrameMatcher::~FrameMatcher() {}

bool FrameMatcher::fusedMatchCall(const cv::Mat &descriptors, const std::vector<Eigen::Vector3f> &features3D,
                             Eigen::Matrix4f &estimatedTransformation, std::vector<cv::DMatch> &inlierMatches,
                             double &pointInlierRatio)
{
    const int n = (int)features3D.size(), np = (int)prevFeatures3D.size();
    if (n > PS_MAX_KPTS || np > PS_MAX_KPTS) return false;
    if ((n > 0 && (descriptors.cols != PS_DESC_BYTES || descriptors.rows != n)) ||
        (np > 0 && (prevDescriptors.cols != PS_DESC_BYTES || prevDescriptors.rows != np)))
        return false; // float descriptors / inconsistent sizes: the generic sequence reports them
    if (!fused_) fused_.reset(new Fused());
    Fused &f = *fused_;
    if (!f.ctx) {
        int dev = 0;
        if (const char *e = std::getenv("PUTSLAM_HIP_DEVICE")) dev = std::atoi(e);
        if (ps_context_create(dev, &f.ctx) != PS_OK) {
            f.ctx = nullptr;
            return false;
        }
    }
    const int need = std::max(std::max(n, np), 1);
    if (!f.stream || f.cap < need) {
        if (f.stream) ps_vo_stream_destroy(f.stream);
        f.stream = nullptr;
        f.cap = std::min(PS_MAX_KPTS, std::max(2048, need + need / 4));
        if (ps_vo_stream_create(f.ctx, f.cap, &f.stream) != PS_OK) {
            f.stream = nullptr;
            return false;
        }
        fusedSynced_ = false;
    }
    RANSAC::parameters rp = matcherParameters.RANSACParams;
    rp.iterationCount = ransacIterations(0.20); // RANSAC.cpp:30
    const PsRansacParams prm = toPs(rp);
    int a = ransacIterations(0.20), b = ransacIterations(rp.minimalInlierRatioThreshold);
    int H = a > b ? a : b;
    H = std::max(1, std::min(H, PS_MAX_HYPOTHESES));
    PsRansacConfig cfg;
    cfg.estimator = PS_EST_RANSAC;
    cfg.numHypotheses = H;
    cfg.seed = seeded_ ? seed_ + (uint64_t)frameCounter : (uint64_t)std::time(nullptr); // RANSAC.cpp:13
    cfg.sampleIdx = nullptr;
    float K[9];
    bool haveK;
    cameraToK(matcherParameters.cameraMatrixMat, K, haveK);
    std::vector<cv::DMatch> m((size_t)f.cap);
    std::vector<uint8_t> mask((size_t)f.cap);
    Eigen::Matrix4f pose = Eigen::Matrix4f::Identity();
    PsRansacStats st;
    int nm = 0;
    if (!fusedSynced_) { // (re)load the previous frame as the resident one
        ps_vo_stream_reset(f.stream);
        int rc = ps_vo_stream_push(f.stream, &prm, &cfg, haveK ? K : nullptr, prevDescriptors.data,
                                   np > 0 ? (size_t)prevDescriptors.step : (size_t)PS_DESC_BYTES,
                                   reinterpret_cast<const float *>(prevFeatures3D.data()), np,
                                   reinterpret_cast<PsDMatch *>(m.data()), &nm, mask.data(), pose.data(), &st);
        if (rc != PS_OK) return false;
    }
    int rc = ps_vo_stream_push(f.stream, &prm, &cfg, haveK ? K : nullptr, descriptors.data,
                               n > 0 ? (size_t)descriptors.step : (size_t)PS_DESC_BYTES,
                               reinterpret_cast<const float *>(features3D.data()), n, reinterpret_cast<PsDMatch *>(m.data()),
                               &nm, mask.data(), pose.data(), &st);
    if (rc != PS_OK || nm < 0) {
        if (rc != PS_OK) std::cerr << "putslam_hip: " << ps_last_error(f.ctx) << std::endl;
        fusedSynced_ = false;
        return false;
    }
    fusedSynced_ = true; // the frame just pushed is the resident "previous" one now
    inlierMatches.clear();
    inlierMatches.reserve((size_t)st.numInliers);
    for (int i = 0; i < nm; ++i)
        if (mask[(size_t)i]) inlierMatches.push_back(m[(size_t)i]);
    estimatedTransformation = pose;
    // RANSAC::pointInlierRatio (RANSAC.h:56-66) was evaluated on the device with two bitmaps over trainIdx
    // (unique inlier train indices / unique matched train indices, 0/0 = NaN like the reference's set sizes)
    pointInlierRatio = st.pointInlierRatio;
    return true;
}

// ---- pipelined form of runVO -------------------------------------------------------------------
void FrameMatcher::setPipeline(int chunkFrames, int lanes)
{
    pipeChunk_ = chunkFrames;
    pipeLanes_ = lanes;
    resetPipeline();
}

void FrameMatcher::resetPipeline()
{
    if (fused_ && fused_->pipe) {
        ps_vo_stream_destroy(fused_->pipe); // (drains; results not dequeued are dropped)
        fused_->pipe = nullptr;
    }
    pipeFirst_ = true;
}

// (re)builds the pipelined stream with room for `cap` keypoints per frame; pair 0 of the new pipeline is pair `pairsSoFar` of
// the sequence (it draws from seed + pairsSoFar, as it would have without the rebuild)
bool FrameMatcher::buildPipeline(int cap, uint64_t pairsSoFar)
{
    Fused &f = *fused_;
    if (f.pipe) {
        ps_vo_stream_destroy(f.pipe);
        f.pipe = nullptr;
    }
    f.pipeCap = cap;
    if (ps_vo_stream_create(f.ctx, f.pipeCap, &f.pipe) != PS_OK) {
        lastError_ = std::string("ps_vo_stream_create: ") + ps_last_error(f.ctx);
        f.pipe = nullptr;
        return false;
    }
    matcherParameters.RANSACParams.errorVersion = matcherParameters.RANSACParams.errorVersionVO; // matcher.cpp:491-492
    RANSAC::parameters rp = matcherParameters.RANSACParams;
    rp.iterationCount = ransacIterations(0.20); // RANSAC.cpp:30
    const PsRansacParams prm = toPs(rp);
    int a = ransacIterations(0.20), b = ransacIterations(rp.minimalInlierRatioThreshold);
    int H = std::max(1, std::min(a > b ? a : b, PS_MAX_HYPOTHESES));
    PsRansacConfig cfg;
    cfg.estimator = PS_EST_RANSAC;
    cfg.numHypotheses = H;
    // pair k of the pipeline draws from seed + k: runVO's seeding (seed_ + frameCounter, the counter starting at 0 with the
    // initial frame)
    if (!pipeSeeded_) {
        pipeSeed_ = seeded_ ? seed_ : (uint64_t)std::time(nullptr); // RANSAC.cpp:13
        pipeSeeded_ = true;
    }
    cfg.seed = pipeSeed_ + pairsSoFar;
    cfg.sampleIdx = nullptr;
    float K[9];
    bool haveK;
    cameraToK(matcherParameters.cameraMatrixMat, K, haveK);
    // what comes back per frame is what Matcher::match returns: the inlier matches in input order + the pose (matcher.cpp:452-516)
    ps_vo_stream_set_result_mode(f.pipe, PS_RESULTS_INLIERS);
    if (ps_vo_stream_configure_async(f.pipe, &prm, &cfg, haveK ? K : nullptr, pipeChunk_, pipeLanes_) != PS_OK) {
        lastError_ = std::string("ps_vo_stream_configure_async: ") + ps_last_error(f.ctx);
        ps_vo_stream_destroy(f.pipe);
        f.pipe = nullptr;
        return false;
    }
    f.pm.resize((size_t)f.pipeCap);
    f.pmask.resize((size_t)f.pipeCap);
    return true;
}

bool FrameMatcher::enqueueFrame(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D)
{
    const int st = enqueueFrameStatus(descriptors, std::move(features3D));
    if (st < 0) std::cerr << "putslam_hip: " << lastError_ << std::endl;
    return st == 1;
}

int FrameMatcher::enqueueFrameStatus(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D)
{
    const int n = (int)features3D.size();
    if (n > PS_MAX_KPTS || (n > 0 && (descriptors.cols != PS_DESC_BYTES || descriptors.rows != n))) {
        lastError_ = "enqueueFrame: descriptors must be n x 32 bytes with n <= PS_MAX_KPTS rows, one per 3-D feature";
        return -1;
    }
    if (!fused_) fused_.reset(new Fused());
    Fused &f = *fused_;
    if (!f.ctx) {
        int dev = 0;
        if (const char *e = std::getenv("PUTSLAM_HIP_DEVICE")) dev = std::atoi(e);
        if (ps_context_create(dev, &f.ctx) != PS_OK) {
            f.ctx = nullptr;
            lastError_ = "ps_context_create failed (no usable HIP device; there is no CPU fallback)";
            return -1;
        }
    }
    if (!f.pipe) {
        // room for a quarter more keypoints than the first frame has; a later frame with more rebuilds the pipeline (below)
        pipeSeeded_ = false;
        if (!buildPipeline(std::min(PS_MAX_KPTS, std::max(2048, n + n / 4)), 0)) return -1;
        pipeFirst_ = true;
    }
    if (n > f.pipeCap) {
        // A frame with more keypoints than the pipeline was built for (runVO's synchronous stream regrows in place; here chunks
        // are in flight).  While results are pending the caller is told "busy" -- the `while (!enqueueFrame) dequeueResult` loop
        // drains them --; with nothing pending the pipeline is rebuilt with more room, the previous frame goes in again as its
        // initial frame and pair numbering (= the hypothesis seeds) continues where it was.
        if (ps_vo_stream_pending(f.pipe) > 0) {
            lastError_ = "enqueueFrame: the frame has more keypoints than the pipeline's capacity; dequeue the pending results, the "
                         "pipeline is rebuilt then";
            return 0;
        }
        const bool hadPrev = !pipeFirst_;
        if (!buildPipeline(std::min(PS_MAX_KPTS, std::max(n + n / 4, 2 * f.pipeCap)), hadPrev ? (uint64_t)frameCounter : 0)) return -1;
        if (hadPrev) {
            const int np = (int)prevFeatures3D.size();
            int rc = ps_vo_stream_push_async(f.pipe, prevDescriptors.data, np > 0 ? (size_t)prevDescriptors.step : (size_t)PS_DESC_BYTES,
                                             reinterpret_cast<const float *>(prevFeatures3D.data()), np);
            if (rc != PS_OK) {
                lastError_ = std::string("ps_vo_stream_push_async (previous frame into the rebuilt pipeline): ") + ps_last_error(f.ctx);
                return -1;
            }
        }
    }
    int rc = ps_vo_stream_push_async(f.pipe, descriptors.data, n > 0 ? (size_t)descriptors.step : (size_t)PS_DESC_BYTES,
                                     reinterpret_cast<const float *>(features3D.data()), n);
    if (rc == PS_ERR_BUSY) {
        lastError_ = "enqueueFrame: the pipeline is full; dequeue results first";
        return 0;
    }
    if (rc != PS_OK) {
        // (the pipelined stream has dropped the chunk this frame belonged to and started a new epoch: the frames enqueued since
        // the last submitted chunk have no results; include/putslam_hip.h)
        lastError_ = std::string("ps_vo_stream_push_async: ") + ps_last_error(f.ctx);
        return -1;
    }
    if (pipeFirst_) {
        frameCounter = 0; // detectInitFeatures, matcher.cpp:17-64
        pipeFirst_ = false;
    } else {
        ++frameCounter;
    }
    features3D.swap(prevFeatures3D); // matcher.cpp:506-513: the enqueued frame is the previous one for whatever comes next
    prevDescriptors = descriptors;
    fusedSynced_ = false;
    return 1;
}

bool FrameMatcher::flushFrames()
{
    if (!fused_ || !fused_->pipe) return true;
    return ps_vo_stream_flush(fused_->pipe) == PS_OK;
}

int FrameMatcher::dequeueResult(Eigen::Matrix4f &estimatedTransformation, std::vector<cv::DMatch> &inlierMatches,
                                double &pointInlierRatio, bool wait)
{
    if (!fused_ || !fused_->pipe) return 0;
    Fused &f = *fused_;
    Eigen::Matrix4f pose = Eigen::Matrix4f::Identity();
    PsRansacStats st;
    int nm = -1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        int rc = ps_vo_stream_pop(f.pipe, wait ? 1 : 0, reinterpret_cast<PsDMatch *>(f.pm.data()), &nm, f.pmask.data(), pose.data(), &st);
        if (rc != PS_OK) {
            std::cerr << "putslam_hip: " << ps_last_error(f.ctx) << std::endl;
            return -1;
        }
        if (nm >= 0 || !wait) break;
        // nothing in flight: frames may still wait in a partly filled chunk
        if (ps_vo_stream_pending(f.pipe) <= 0 || ps_vo_stream_flush(f.pipe) != PS_OK) break;
    }
    if (nm < 0) return 0;
    inlierMatches.assign(f.pm.begin(), f.pm.begin() + nm); // (PS_RESULTS_INLIERS: the popped matches ARE the inliers)
    estimatedTransformation = pose;
    pointInlierRatio = st.pointInlierRatio;
    return 1;
}

double FrameMatcher::match(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D, Eigen::Matrix4f &estimatedTransformation,
                      std::vector<cv::DMatch> &inlierMatches)
{
    matcherParameters.RANSACParams.errorVersion = matcherParameters.RANSACParams.errorVersionVO; // matcher.cpp:491-492
    double ratio = 0.0;
    if (!(fusedMatch_ && fusedMatchCall(descriptors, features3D, estimatedTransformation, inlierMatches, ratio))) {
        std::vector<cv::DMatch> matches = performMatching(prevDescriptors, descriptors); // matcher.cpp:470
        RANSAC ransac(matcherParameters.RANSACParams, matcherParameters.cameraMatrixMat);
        if (seeded_) ransac.setSampleSeed(seed_ + (uint64_t)frameCounter);
        estimatedTransformation = ransac.estimateTransformation(prevFeatures3D, features3D, matches, inlierMatches);
        fusedSynced_ = false;
        ratio = RANSAC::pointInlierRatio(inlierMatches, matches); // :515
    }
    features3D.swap(prevFeatures3D); // :506-513 save computed values for the next iteration
    prevDescriptors = descriptors;
    ++frameCounter;
    return ratio;
}

double FrameMatcher::matchFeatureLoopClosure(cv::Mat desc0, std::vector<Eigen::Vector3f> pts0, cv::Mat desc1,
                                        std::vector<Eigen::Vector3f> pts1, Eigen::Matrix4f &estimatedTransformation,
                                        std::vector<cv::DMatch> &inlierMatches)
{
    if (pts0.size() < 10 || pts1.size() < 10) return 0; // matcher.cpp:830-834
    std::vector<cv::DMatch> matches = performMatching(desc0, desc1); // :835
    if (matches.size() <= 0) return -1.0;                            // :838-839
    matcherParameters.RANSACParams.errorVersion = matcherParameters.RANSACParams.errorVersionMap; // :843-844
    RANSAC ransac(matcherParameters.RANSACParams, matcherParameters.cameraMatrixMat);
    if (seeded_) ransac.setSampleSeed(seed_ + 0x9E3779B97F4A7C15ull + (uint64_t)frameCounter);
    estimatedTransformation = ransac.estimateTransformation(pts0, pts1, matches, inlierMatches);
    return RANSAC::pointInlierRatio(inlierMatches, matches);
}

double FrameMatcher::matchXYZ(const std::vector<MapFeatureXYZ> &mapFeatures, cv::Mat currentPoseDescriptors,
                         std::vector<Eigen::Vector3f> &currentPoseFeatures3D, const std::vector<int> &currentPoseOctaves,
                         const std::vector<double> &currentPoseDetDists, Eigen::Matrix4f &estimatedTransformation,
                         std::vector<cv::DMatch> &inlierMatches, int computationNumber)
{
    double matchingXYZSphereRadius = matcherParameters.OpenCVParams.matchingXYZSphereRadius; // matcher.cpp:616-622
    double matchingXYZacceptRatioOfBestMatch = matcherParameters.OpenCVParams.matchingXYZacceptRatioOfBestMatch;
    if (computationNumber > 1) {
        matchingXYZSphereRadius += 0.02 * (computationNumber - 1);
        matchingXYZacceptRatioOfBestMatch = std::max(0.1, matchingXYZacceptRatioOfBestMatch - 0.05 * (computationNumber - 1));
    }
    const size_t nmap = mapFeatures.size(), ncur = currentPoseFeatures3D.size();
    std::vector<int32_t> curLevels(ncur), mapLevels(nmap);
    for (size_t i = 0; i < ncur; ++i) { // :639-652; curDist = Eigen float norm()
        const Eigen::Vector3f &p = currentPoseFeatures3D[i];
        float nrm = std::sqrt(p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]));
        curLevels[i] = ps_predicted_level(currentPoseOctaves[i], currentPoseDetDists[i], (double)nrm);
    }
    std::vector<Eigen::Vector3f> mapFeaturePositions3D(nmap);
    std::vector<unsigned char> mapDesc(nmap * 32);
    for (size_t j = 0; j < nmap; ++j) { // :681-692,701-702
        const MapFeatureXYZ &f = mapFeatures[j];
        double curDist = std::sqrt(f.position[0] * f.position[0] + f.position[1] * f.position[1] + f.position[2] * f.position[2]);
        mapLevels[j] = ps_predicted_level(f.octave, f.detDist, curDist);
        mapFeaturePositions3D[j] = Eigen::Vector3f((float)f.position[0], (float)f.position[1], (float)f.position[2]);
        if (!f.descriptor.empty()) std::memcpy(&mapDesc[j * 32], f.descriptor.data, 32);
    }
    std::vector<cv::DMatch> matches;
    int status;
    PsContext *ctx = threadContext(&status);
    if (!ctx || nmap == 0 || ncur == 0) return -1.0;
    int cap = (int)(4 * nmap + 16), n = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        matches.resize((size_t)cap);
        status = ps_match_xyz(ctx, reinterpret_cast<const float *>(mapFeaturePositions3D.data()), mapDesc.data(), 32,
                              mapLevels.data(), (int)nmap, reinterpret_cast<const float *>(currentPoseFeatures3D.data()),
                              currentPoseDescriptors.data, (size_t)currentPoseDescriptors.step, curLevels.data(), (int)ncur,
                              matchingXYZSphereRadius, matchingXYZacceptRatioOfBestMatch,
                              reinterpret_cast<PsDMatch *>(matches.data()), cap, &n);
        if (status == PS_OK || n <= cap) break;
        cap = n;
    }
    if (status != PS_OK) {
        std::cerr << "putslam_hip: " << ps_last_error(ctx) << std::endl;
        return -1.0;
    }
    matches.resize((size_t)n);
    if (matcherParameters.verbose > 0) std::cout << "MatchesXYZ - we found : " << matches.size() << std::endl;
    if (matches.size() <= 0) return -1.0; // :755-756
    matcherParameters.RANSACParams.errorVersion = matcherParameters.RANSACParams.errorVersionMap; // :760-761
    RANSAC ransac(matcherParameters.RANSACParams, matcherParameters.cameraMatrixMat);
    if (seeded_) ransac.setSampleSeed(seed_ + 0x51ED270B0B5ull + (uint64_t)frameCounter);
    estimatedTransformation = ransac.estimateTransformation(mapFeaturePositions3D, currentPoseFeatures3D, matches, inlierMatches);
    return RANSAC::pointInlierRatio(inlierMatches, matches);
}

void VOTrajectory::addIncrement(Eigen::Matrix4f inc)
{
    // PUTSLAM.cpp:735-740
    double translationVO = std::sqrt(std::pow(inc(0, 3), 2) + std::pow(inc(1, 3), 2) + std::pow(inc(2, 3), 2));
    if (translationVO > 0.1) inc = Eigen::Matrix4f::Identity();
    Eigen::Matrix4f r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float p0 = VOPoseEstimate(i, 0) * inc(0, j), p1 = VOPoseEstimate(i, 1) * inc(1, j);
            float p2 = VOPoseEstimate(i, 2) * inc(2, j), p3 = VOPoseEstimate(i, 3) * inc(3, j);
            r(i, j) = (p0 + p1) + (p2 + p3);
        }
    VOPoseEstimate = r;
}

std::string VOTrajectory::freiburgLine(const Eigen::Matrix4f &T, double timestamp)
{
    // saveTrajectoryFreiburgFormat, PUTSLAM.cpp:1006-1016; Eigen::Quaternion<float>(Matrix3f)
    float m[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = T(i, j);
    float q[4]; // x y z w
    float t = (m[0][0] + m[1][1]) + m[2][2];
    if (t > 0.0f) {
        t = std::sqrt(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (m[2][1] - m[1][2]) * t;
        q[1] = (m[0][2] - m[2][0]) * t;
        q[2] = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(((m[i][i] - m[j][j]) - m[k][k]) + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (m[k][j] - m[j][k]) * t;
        q[j] = (m[j][i] + m[i][j]) * t;
        q[k] = (m[k][i] + m[i][k]) * t;
    }
    std::ostringstream ossTimestamp;
    ossTimestamp << std::setfill('0') << std::setprecision(17) << timestamp;
    std::ostringstream o;
    o << ossTimestamp.str() << " " << T(0, 3) << " " << T(1, 3) << " " << T(2, 3) << " " << q[0] << " " << q[1] << " " << q[2]
      << " " << q[3];
    return o.str();
}

// "A single instance of OpenCV matcher", matcherOpenCV.cpp:20 -- the same ownership rule for the hot-path matcher
FrameMatcherHIP::Ptr matcherClass, loopClosingMatcherClass;

FrameMatcher *createFrameMatcher(void)
{
    matcherClass.reset(new FrameMatcherHIP());
    return matcherClass.get();
}
FrameMatcher *createFrameMatcher(const std::string _parametersFile, const std::string _grabberParametersFile)
{
    matcherClass.reset(new FrameMatcherHIP(_parametersFile, _grabberParametersFile));
    return matcherClass.get();
}
FrameMatcher *createLoopClosingFrameMatcher(const std::string _parametersFile, const std::string _grabberParametersFile)
{
    loopClosingMatcherClass.reset(new FrameMatcherHIP(_parametersFile, _grabberParametersFile));
    return loopClosingMatcherClass.get();
}

// ---------------------------------------------------------------------------------------------
FrameMatcherHIP::FrameMatcherHIP(void) : FrameMatcher("OpenCV Matcher") { fusedMatch_ = true; }
// XML parsing (tinyXML, matcher.h:188-357) is outside the path: parameters are plain members to set.
FrameMatcherHIP::FrameMatcherHIP(const std::string, const std::string) : FrameMatcher("OpenCVMatcher") { fusedMatch_ = true; }
FrameMatcherHIP::~FrameMatcherHIP(void) {}
const std::string &FrameMatcherHIP::getName() const { return name; }

std::vector<cv::DMatch> FrameMatcherHIP::performMatching(cv::Mat prevDescriptors, cv::Mat descriptors)
{
    return hammingCrossCheckMatch(prevDescriptors, descriptors);
}

// MatcherOpenCV::performMatching (matcherOpenCV.cpp:198-206) as a free function: the body the reference's own class
// gets in a PUTSLAM build.
std::vector<cv::DMatch> hammingCrossCheckMatch(cv::Mat prevDescriptors, cv::Mat descriptors)
{
    std::vector<cv::DMatch> matches;
    if (prevDescriptors.empty() || descriptors.empty()) return matches;
    if (prevDescriptors.cols != PS_DESC_BYTES || descriptors.cols != PS_DESC_BYTES) {
        // float descriptors (SURF/SIFT, NORM_L2: matcherOpenCV.cpp:100-102) are outside the path
        std::cerr << "putslam_hip: performMatching supports 32-byte binary descriptors (ORB/LDB) only" << std::endl;
        return matches;
    }
    int status;
    PsContext *ctx = threadContext(&status);
    if (!ctx) return matches;
    matches.resize((size_t)prevDescriptors.rows);
    int n = 0;
    status = ps_match_hamming256(ctx, prevDescriptors.data, prevDescriptors.rows, (size_t)prevDescriptors.step, descriptors.data,
                                 descriptors.rows, (size_t)descriptors.step, reinterpret_cast<PsDMatch *>(matches.data()), &n);
    if (status != PS_OK) {
        std::cerr << "putslam_hip: " << ps_last_error(ctx) << std::endl;
        n = 0;
    }
    matches.resize((size_t)n);
    return matches;
}

} // namespace putslam_hip
