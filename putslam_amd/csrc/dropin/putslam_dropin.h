// putslam_dropin.h -- the reference's C++ surface for the hot path, implemented over the C ABI
// (include/putslam_hip.h).  Class names, namespaces, signatures, ownership and error behaviour follow
// the reference so that PUTSLAM's callers compile unchanged against this header:
//
//   ::RANSAC                       include/putslam/TransformEst/RANSAC.h:20-66, src/TransformEst/RANSAC.cpp
//   ::RANSAC_USAC                  include/putslam/USAC/USAC_wrapper.h:16-65, src/USAC/USAC_wrapper.cpp
//   putslam::TransformEst          include/putslam/TransformEst/transformEst.h:16-26
//   putslam::KabschEst + factory   include/putslam/TransformEst/kabschEst.h, src/TransformEst/kabschEst.cpp
//
// The Matcher plugin itself (putslam::Matcher / ::MatcherOpenCV, include/putslam/Matcher/matcher.h:24,100-151,405-422)
// keeps ITS OWN class in a PUTSLAM build: detection, description and tracking are image-domain OpenCV stages outside the
// path.  This library therefore defines no symbol of that name.  What it provides for the plugin is
//   putslam_hip::FrameMatcher / FrameMatcherHIP   the hot-path state machine behind Matcher::match / runVO /
//                                  matchXYZ / matchFeatureLoopClosure (matcher.cpp:452-516,606-861) on descriptors and
//                                  3-D points, which the reference's methods call after their detect / describe part;
//   putslam_matcher_glue.h         the same entry points with the reference's own argument types (SensorFrame-derived
//                                  descriptors, std::vector<MapFeature>, framesIds[2], pairedFeatures), as templates;
//   INTEGRATION.md section 2       the bodies a maintainer puts into matcher.cpp / matcherOpenCV.cpp.
//   RGBD helpers                   include/putslam/RGBD/RGBD.h:38-51, src/RGBD/RGBD.cpp:10-16,30-65,92-98
//
// Everything computes on the GPU through libputslam_hip.so; there is no host fallback.
#pragma once

#include <cstdint>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "putslam_compat_types.h"

#define PUTSLAM_HIP_DESC_BYTES 32 // ORB / LDB rows (matcherOpenCV.cpp:90, ldb.cpp:61,657)

// ---------------------------------------------------------------------------------------------
class RANSAC {
  public:
    enum ERROR_VERSION { EUCLIDEAN_ERROR, REPROJECTION_ERROR, EUCLIDEAN_AND_REPROJECTION_ERROR, MAHALANOBIS_ERROR, ADAPTIVE_ERROR };
    struct parameters {
        int verbose;
        int errorVersion, errorVersionVO, errorVersionMap;
        double inlierThresholdEuclidean, inlierThresholdReprojection, inlierThresholdMahalanobis;
        double minimalInlierRatioThreshold;
        int minimalNumberOfMatches;
        int usedPairs;
        int iterationCount;
    };

    RANSAC(RANSAC::parameters RANSACParameters, cv::Mat cameraMatrix = cv::Mat());

    // prevFeatures / features: 3-D points of the previous / current frame; matches: queryIdx into
    // prevFeatures, trainIdx into features.  Returns the pose mapping current-frame points into the
    // previous frame; identity + cleared inliers on failure (no exceptions, no error codes).
    Eigen::Matrix4f estimateTransformation(std::vector<Eigen::Vector3f> prevFeatures, std::vector<Eigen::Vector3f> features,
                                           std::vector<cv::DMatch> matches, std::vector<cv::DMatch> &inlierMatches);

    static double pointInlierRatio(std::vector<cv::DMatch> &inlierMatches, std::vector<cv::DMatch> &allMatches)
    {
        std::set<int> inlier, all;
        for (auto &m : allMatches) all.insert(m.trainIdx);
        for (auto &in : inlierMatches) inlier.insert(in.trainIdx);
        return double(inlier.size()) / double(all.size());
    }

    // --- additions with no reference counterpart (the reference seeds rand() from time(0), RANSAC.cpp:13) ---
    void setSampleSeed(uint64_t seed) { seed_ = seed; }
    uint64_t sampleSeed() const { return seed_; }
    int lastStatus() const { return lastStatus_; }   // PsStatus of the last call (0 = ok)

  private:
    cv::Mat cameraMatrix;
    parameters RANSACParams;
    uint64_t seed_;
    int lastStatus_ = 0;
};

// ---------------------------------------------------------------------------------------------
class PUTSLAMEstimator { // only the parameter block of the reference class is part of the surface
  public:
    struct parameters {
        int verbose;
        int errorVersion, errorVersionVO, errorVersionMap;
        double inlierThresholdEuclidean, inlierThresholdReprojection, inlierThresholdMahalanobis;
        double minimalInlierRatioThreshold;
        int usedPairs;
        int iterationCount;
    };
};

class RANSAC_USAC {
  public:
    RANSAC_USAC(PUTSLAMEstimator::parameters RANSACParameters, cv::Mat cameraMatrix = cv::Mat());
    ~RANSAC_USAC();
    Eigen::Matrix4f estimateTransformation(std::vector<Eigen::Vector3f> prevFeatures, std::vector<Eigen::Vector3f> features,
                                           std::vector<cv::DMatch> matches, std::vector<cv::DMatch> &bestInlierMatches);
    void setSampleSeed(uint64_t seed) { seed_ = seed; }
    // samples made available to the USAC loop; default = the reference's usac_max_hypotheses_ (USAC_wrapper.cpp:70): the loop
    // stops by its own criterion long before, and a long cap costs 10 us per call (rounds 1-3 defaulted to 4096, which cut
    // the schedule short below 10 % inliers)
    void setMaxHypotheses(int h) { maxHyp_ = h; }
    int lastStatus() const { return lastStatus_; }

  private:
    cv::Mat cameraMatrix;
    PUTSLAMEstimator::parameters params_;
    uint64_t seed_;
    int maxHyp_ = 850000;
    int lastStatus_ = 0;
};

// ---------------------------------------------------------------------------------------------
class RGBD {
  public:
    static int roundSize(double x, int size);
    static std::vector<Eigen::Vector3f> keypoints2Dto3D(std::vector<cv::Point2f> undistortedFeatures2D, cv::Mat depthImage,
                                                        cv::Mat cameraMatrix, double depthImageScale = 5000, int startingID = 0);
    static std::vector<cv::Point2f> points3Dto2D(std::vector<Eigen::Vector3f> features3D, cv::Mat cameraMatrix);
    // distCoeffs: 1x5 (or 5x1) CV_32F/CV_64F-like access through at<float>: (k1, k2, p1, p2, k3)
    static std::vector<cv::Point2f> removeImageDistortion(std::vector<cv::Point2f> &features, cv::Mat cameraMatrix,
                                                          cv::Mat distCoeffs);
};

namespace putslam {

// ---------------------------------------------------------------------------------------------
class TransformEst {
  public:
    virtual const std::string &getName() const = 0;
    // setA, setB: N x 3; returns a reference to the member `transformation` (maps A onto B)
    virtual Mat34 &computeTransformation(const Eigen::MatrixXd &setA, const Eigen::MatrixXd &setB) = 0;
    virtual ~TransformEst() {}

  protected:
    Mat34 transformation;
};

class KabschEst : public TransformEst {
  public:
    typedef std::unique_ptr<KabschEst> Ptr;
    KabschEst(void);
    const std::string &getName() const;
    Mat34 &computeTransformation(const Eigen::MatrixXd &setA, const Eigen::MatrixXd &setB);
    virtual ~KabschEst() {}

  private:
    const std::string name;
};

// library-owned singleton, raw pointer returned, a second call replaces the instance (kabschEst.cpp:8,70-73)
TransformEst *createKabschEstimator(void);

} // namespace putslam

namespace putslam_hip {

// ---------------------------------------------------------------------------------------------
// Hot-path part of the Matcher plugin.  Detection / description / tracking are image-domain OpenCV
// stages outside the path: their pure virtuals are not part of this class; the
// frame enters at the point where Matcher::match has descriptors and 3-D points (matcher.cpp:467-480).
// Deliberately NOT named putslam::Matcher: the reference's class of that name stays in a PUTSLAM build and
// delegates to this one, so the two can be linked into one program.
class FrameMatcher {
  public:
    struct MatcherParameters {
        int verbose = 0;
        RANSAC::parameters RANSACParams;
        cv::Mat cameraMatrixMat; // 3x3 CV_32FC1
        struct {
            double matchingXYZSphereRadius = 0.12;            // putslammatcherOpenCVParameters.xml:71
            double matchingXYZacceptRatioOfBestMatch = 0.55;  // :72
        } OpenCVParams;
        MatcherParameters();
    };

    // What Matcher::matchXYZ reads of a MapFeature (putslam_defs.h:120-216): position, the descriptor of the
    // chosen view with its octave and detection distance (ExtendedDescriptor), and the feature id.
    struct MapFeatureXYZ {
        unsigned int id = 0;
        double position[3] = {0, 0, 0};
        cv::Mat descriptor; // 1 x 32 CV_8U
        int octave = 0;
        double detDist = 1.0;
    };
    static constexpr double scaleFactor = 1.2; // matcher.h:26-27
    static constexpr int nLevels = 8;

    // Matcher::featureSet (matcher.h:31-37) restricted to what the hot path holds: the image-domain members
    // (cv::KeyPoint lists, 2-D positions, detection distances) belong to the out-of-scope detect/describe stages.
    struct featureSet {
        cv::Mat descriptors;
        std::vector<Eigen::Vector3f> feature3D;
    };

    FrameMatcher(const std::string _name);
    virtual ~FrameMatcher();
    // one resident frame store (context + stream + HBM slots) per instance: not copyable
    FrameMatcher(const FrameMatcher &) = delete;
    FrameMatcher &operator=(const FrameMatcher &) = delete;
    virtual const std::string &getName() const = 0;
    virtual std::vector<cv::DMatch> performMatching(cv::Mat prevDescriptors, cv::Mat descriptors) = 0;

    // First frame: stores descriptors + 3-D points as the "previous" state (detectInitFeatures, matcher.cpp:17-64).
    void detectInitFeatures(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D);
    // matcher.cpp:470-515 from the described frame on: performMatching(prev, cur) -> RANSAC with errorVersionVO
    // -> swap cur -> prev; returns RANSAC::pointInlierRatio(inliers, matches).
    double match(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D, Eigen::Matrix4f &estimatedTransformation,
                 std::vector<cv::DMatch> &inlierMatches);
    double runVO(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D, Eigen::Matrix4f &estimatedTransformation,
                 std::vector<cv::DMatch> &inlierMatches)
    {
        return match(descriptors, features3D, estimatedTransformation, inlierMatches);
    }
    // Loop-closure matching of two feature sets (matchFeatureLoopClosure, matcher.cpp:802-861): BF match +
    // RANSAC with errorVersionMap; returns -1 when there are no matches, 0 when either set has < 10 features.
    double matchFeatureLoopClosure(cv::Mat desc0, std::vector<Eigen::Vector3f> pts0, cv::Mat desc1,
                                   std::vector<Eigen::Vector3f> pts1, Eigen::Matrix4f &estimatedTransformation,
                                   std::vector<cv::DMatch> &inlierMatches);
    // Guided matching of map features against the current pose's keypoints + RANSAC with errorVersionMap
    // (matchXYZ, matcher.cpp:606-798): returns RANSAC::pointInlierRatio, or -1 when nothing matched.
    double matchXYZ(const std::vector<MapFeatureXYZ> &mapFeatures, cv::Mat currentPoseDescriptors,
                    std::vector<Eigen::Vector3f> &currentPoseFeatures3D, const std::vector<int> &currentPoseOctaves,
                    const std::vector<double> &currentPoseDetDists, Eigen::Matrix4f &estimatedTransformation,
                    std::vector<cv::DMatch> &inlierMatches, int computationNumber = 1);
    // ---- pipelined form of runVO (ps_vo_stream_push_async / ps_vo_stream_pop, include/putslam_hip.h): the same call shape --
    // one frame per call, the previous frame kept as state (matcher.cpp:452-516) -- with the result returned with a LAG, so that
    // uploads, kernels and downloads of consecutive frames overlap (BASELINE configs[2]: a sequence streamed through the matcher).
    // Results are those runVO returns for the same frames in the same order (pair k draws from seed + k either way).
    //   setPipeline    frames per submitted chunk (1 = lowest latency) and chunks in flight; before the first enqueueFrame
    //   enqueueFrame   the first frame after construction / resetPipeline is the initial one (detectInitFeatures); returns false
    //                  when the frame was NOT taken.  enqueueFrameStatus tells why: 1 = taken, 0 = busy -- the pipeline is full, or
    //                  the frame has more keypoints than the pipeline was built for and results are still pending: dequeue
    //                  results and call again (the `while (!enqueueFrame(..)) dequeueResult(..)` loop does; once drained the
    //                  pipeline is rebuilt with more room and numbering continues) --, -1 = error (lastError() has the text;
    //                  enqueueFrame also prints it): retrying the same frame will not help
    //   dequeueResult  1 = the next frame's result (in frame order), 0 = none ready (wait = false) or none pending,
    //                  -1 = error; a partly filled chunk is submitted when a waiting call finds nothing in flight
    //   flushFrames    submits a partly filled chunk now
    void setPipeline(int chunkFrames, int lanes);
    bool enqueueFrame(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D);
    int enqueueFrameStatus(cv::Mat descriptors, std::vector<Eigen::Vector3f> features3D);
    const std::string &lastError() const { return lastError_; }
    int dequeueResult(Eigen::Matrix4f &estimatedTransformation, std::vector<cv::DMatch> &inlierMatches, double &pointInlierRatio,
                      bool wait = true);
    bool flushFrames();
    void resetPipeline();
    int getNumberOfFeatures() const { return (int)prevFeatures3D.size(); }
    featureSet getFeatures() // matcher.cpp:980-989
    {
        featureSet r;
        r.descriptors = prevDescriptors;
        r.feature3D = prevFeatures3D;
        return r;
    }
    void setSampleSeed(uint64_t s) { seed_ = s; seeded_ = true; }

    MatcherParameters matcherParameters;

  protected:
    const std::string name;
    cv::Mat prevDescriptors;
    std::vector<Eigen::Vector3f> prevFeatures3D;
    int frameCounter;
    uint64_t seed_ = 0;
    bool seeded_ = false;
    // MatcherOpenCV sets fusedMatch_: match() then runs performMatching + RANSAC as ONE call against the previous
    // frame kept in HBM (ps_vo_stream_push: the prevDescriptors / prevFeatures3D state of matcher.h:379-384 lives on
    // the GPU as well, one upload and one download per frame) -- same results as the two separate calls.  A class
    // that overrides performMatching leaves it false and gets the generic sequence.
    bool fusedMatch_ = false;
    struct Fused;
    std::unique_ptr<Fused> fused_;
    bool fusedSynced_ = false; // the resident frame is prevDescriptors / prevFeatures3D
    int pipeChunk_ = 32, pipeLanes_ = 0; // (lanes 0 = the library's choice by chunk size)
    bool pipeFirst_ = true;    // the next enqueued frame has no predecessor
    uint64_t pipeSeed_ = 0;    // hypothesis seed of the pipeline's pair 0 (pair k draws from pipeSeed_ + k, across rebuilds)
    bool pipeSeeded_ = false;
    std::string lastError_;
    bool buildPipeline(int cap, uint64_t pairsSoFar);
    bool fusedMatchCall(const cv::Mat &descriptors, const std::vector<Eigen::Vector3f> &features3D,
                        Eigen::Matrix4f &estimatedTransformation, std::vector<cv::DMatch> &inlierMatches,
                        double &pointInlierRatio);
};

// MatcherOpenCV::performMatching (matcherOpenCV.cpp:198-206) as a free function over the calling thread's context.
std::vector<cv::DMatch> hammingCrossCheckMatch(cv::Mat prevDescriptors, cv::Mat descriptors);

// Library-owned singletons with the reference factories' ownership rules (raw pointer returned, a second call
// replaces the instance: matcherOpenCV.cpp:20-47): one for the VO thread, one for the loop-closure thread.
FrameMatcher *createFrameMatcher(void);
FrameMatcher *createFrameMatcher(const std::string _parametersFile, const std::string _grabberParametersFile);
FrameMatcher *createLoopClosingFrameMatcher(const std::string _parametersFile, const std::string _grabberParametersFile);

// ---------------------------------------------------------------------------------------------
// VO driver (PUTSLAM::startProcessing, src/PUTSLAM/PUTSLAM.cpp:733-740,1006-1016): pose composition with
// the 0.1 m gate and the TUM trajectory line format.
struct VOTrajectory {
    Eigen::Matrix4f VOPoseEstimate = Eigen::Matrix4f::Identity();
    void addIncrement(Eigen::Matrix4f poseIncrement);
    static std::string freiburgLine(const Eigen::Matrix4f &pose, double timestamp);
};

// The concrete matcher: performMatching = cv::BFMatcher(NORM_HAMMING, crossCheck = true).match(prev, cur) on the GPU
// (the body MatcherOpenCV::performMatching gets in a PUTSLAM build, matcherOpenCV.cpp:198-206), match() fused with the
// RANSAC against the frame resident in HBM.
class FrameMatcherHIP : public FrameMatcher {
  public:
    typedef std::unique_ptr<FrameMatcherHIP> Ptr;
    FrameMatcherHIP(void);
    FrameMatcherHIP(const std::string _parametersFile, const std::string _grabberParametersFile);
    ~FrameMatcherHIP(void);
    virtual const std::string &getName() const;
    virtual std::vector<cv::DMatch> performMatching(cv::Mat prevDescriptors, cv::Mat descriptors);
};

} // namespace putslam_hip
