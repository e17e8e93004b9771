// test_reference_shaped.cpp -- a translation unit shaped like the reference's Matcher plugin, linked against the drop-in.
//
// What it proves ("Fix the Matcher boundary"):
//  (1) a program may contain classes named putslam::Matcher / ::MatcherOpenCV and the factories createMatcherOpenCV /
//      createloopClosingMatcherOpenCV with the reference's signatures (include/putslam/Matcher/matcher.h:24,100-151,
//      405-422; matcherOpenCV.h:17-30) AND link libputslam_dropin.so: the library defines no symbol of those names;
//  (2) with the bodies of INTEGRATION.md section 2 (detect / describe are stubs here: out of scope) those methods give
//      the oracle's results: performMatching, match / runVO over a sequence, matchFeatureLoopClosure with the
//      reference's argument types (std::vector<MapFeature>[2], int[2], pairedFeatures);
//  (3) N3: loop-closure matching on a second instance and a second thread, concurrently with the VO
//      matcher, is bit-identical to the oracle (pairs, pose, point-inlier ratio), and the 0 / -1 returns hold.
// The data types below have the shape of include/putslam/Defs/putslam_defs.h:120-216 (names and members only).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <thread>

#include "putslam_matcher_glue.h"

namespace putslam {

struct Vec3 { // stands for Eigen::Translation<double,3> (putslam_defs.h:31)
    double c[3];
    Vec3() : c{0, 0, 0} {}
    Vec3(double x, double y, double z) : c{x, y, z} {}
    double x() const { return c[0]; }
    double y() const { return c[1]; }
    double z() const { return c[2]; }
};

class ExtendedDescriptor {
  public:
    cv::Point2f point2D, point2DUndist;
    Vec3 point3D;
    cv::Mat descriptor;
    int octave = 0;
    double detDist = 0;
    ExtendedDescriptor() {}
    ExtendedDescriptor(cv::Point2f _point2D, cv::Point2f _point2DUndist, Vec3 _point3D, cv::Mat _descriptor, int _octave,
                       double _detDist)
        : point2D(_point2D), point2DUndist(_point2DUndist), point3D(_point3D), descriptor(_descriptor), octave(_octave),
          detDist(_detDist)
    {
    }
};

class MapFeature {
  public:
    Vec3 position;
    std::map<unsigned int, ExtendedDescriptor> descriptors;
    double u = 0, v = 0;
    unsigned int id = 0;
    std::vector<unsigned int> posesIds;
};

class SensorFrame { // what the stub front end below reads: the frame index stands for the images
  public:
    cv::Mat rgbImage, depthImage;
    double timestamp = 0;
    double depthImageScale = 5000;
    int index = 0;
};

// Reference-shaped plugin base class: same public / virtual surface, bodies as in INTEGRATION.md section 2.
class Matcher {
  public:
    struct MatcherParameters {
        int verbose = 0;
        RANSAC::parameters RANSACParams;
        cv::Mat cameraMatrixMat;
    };
    Matcher(const std::string _name) : name(_name), frame_id(0) {}
    virtual ~Matcher() {}
    virtual const std::string &getName() const = 0;

    void detectInitFeatures(const SensorFrame &sensorData)
    {
        putslam_hip::detectInitFeatures(hot(), sensorData, frontEnd());
    }
    double runVO(const SensorFrame &currentSensorFrame, Eigen::Matrix4f &estimatedTransformation,
                 std::vector<cv::DMatch> &inlierMatches)
    {
        return match(currentSensorFrame, estimatedTransformation, inlierMatches);
    }
    double match(const SensorFrame &sensorData, Eigen::Matrix4f &estimatedTransformation,
                 std::vector<cv::DMatch> &foundInlierMatches)
    {
        return putslam_hip::match(hot(), sensorData, frontEnd(), estimatedTransformation, foundInlierMatches);
    }
    double matchFeatureLoopClosure(std::vector<MapFeature> featureSets[2], int framesIds[2],
                                   std::vector<std::pair<int, int>> &pairedFeatures, Eigen::Matrix4f &estimatedTransformation)
    {
        return putslam_hip::matchFeatureLoopClosure(hot(), featureSets, framesIds, pairedFeatures, estimatedTransformation);
    }
    int getNumberOfFeatures() { return hot().getNumberOfFeatures(); }

    MatcherParameters matcherParameters;
    // test hooks (no reference counterpart): the frames the stub front end hands out, and the sample seed
    const std::vector<cv::Mat> *frameDescriptors = nullptr;
    const std::vector<std::vector<Eigen::Vector3f>> *framePoints = nullptr;
    void setSampleSeed(uint64_t s) { hot().setSampleSeed(s); }
    void syncParameters() // the XML loader's job in the reference (matcher.h:188-357)
    {
        hot().matcherParameters.RANSACParams = matcherParameters.RANSACParams;
        hot().matcherParameters.cameraMatrixMat = matcherParameters.cameraMatrixMat;
    }

  protected:
    const std::string name;
    int frame_id;
    virtual std::vector<cv::KeyPoint> detectFeatures(cv::Mat rgbImage) = 0;
    virtual cv::Mat describeFeatures(cv::Mat rgbImage, std::vector<cv::KeyPoint> &features) = 0;
    virtual std::vector<cv::DMatch> performMatching(cv::Mat prevDescriptors, cv::Mat descriptors) = 0;
    virtual std::vector<cv::DMatch> performTracking(cv::Mat prevImg, cv::Mat img, std::vector<cv::Point2f> &prevFeatures,
                                                    std::vector<cv::Point2f> &features) = 0;

  private:
    // the hot-path state machine this plugin instance delegates to (one per instance: VO thread, loop-closure thread)
    putslam_hip::FrameMatcherHIP hot_;
    putslam_hip::FrameMatcher &hot() { return hot_; }
    // stands for detectFeatures -> describeFeatures -> removeImageDistortion -> keypoints2Dto3D (matcher.cpp:457-480)
    std::function<void(const SensorFrame &, cv::Mat &, std::vector<Eigen::Vector3f> &)> frontEnd()
    {
        return [this](const SensorFrame &f, cv::Mat &descriptors, std::vector<Eigen::Vector3f> &features3D) {
            std::vector<cv::KeyPoint> kp = detectFeatures(f.rgbImage);
            (void)describeFeatures(f.rgbImage, kp);
            descriptors = (*frameDescriptors)[(size_t)f.index];
            features3D = (*framePoints)[(size_t)f.index];
        };
    }
};

Matcher *createMatcherOpenCV(void);
Matcher *createloopClosingMatcherOpenCV(const std::string _parametersFile, const std::string _grabberParametersFile);

} // namespace putslam

// The reference declares its concrete matcher in the global namespace (matcherOpenCV.h:26-30).
class MatcherOpenCV : public putslam::Matcher {
  public:
    typedef std::unique_ptr<MatcherOpenCV> Ptr;
    MatcherOpenCV(void) : putslam::Matcher("OpenCV Matcher") {}
    MatcherOpenCV(const std::string, const std::string) : putslam::Matcher("OpenCVMatcher") {}
    virtual const std::string &getName() const { return name; }
    std::vector<cv::DMatch> publicPerformMatching(cv::Mat a, cv::Mat b) { return performMatching(a, b); }

  private:
    virtual std::vector<cv::KeyPoint> detectFeatures(cv::Mat) { return std::vector<cv::KeyPoint>(); }                 // stub
    virtual cv::Mat describeFeatures(cv::Mat, std::vector<cv::KeyPoint> &) { return cv::Mat(); }                       // stub
    virtual std::vector<cv::DMatch> performTracking(cv::Mat, cv::Mat, std::vector<cv::Point2f> &, std::vector<cv::Point2f> &)
    {
        return std::vector<cv::DMatch>();                                                                               // stub
    }
    // the whole body of MatcherOpenCV::performMatching (matcherOpenCV.cpp:198-206)
    virtual std::vector<cv::DMatch> performMatching(cv::Mat prevDescriptors, cv::Mat descriptors)
    {
        return putslam_hip::hammingCrossCheckMatch(prevDescriptors, descriptors);
    }
};

namespace putslam {
MatcherOpenCV::Ptr matcherClass, loopClosingMatcherClass; // matcherOpenCV.cpp:20
Matcher *createMatcherOpenCV(void)
{
    matcherClass.reset(new MatcherOpenCV());
    return matcherClass.get();
}
Matcher *createloopClosingMatcherOpenCV(const std::string a, const std::string b)
{
    loopClosingMatcherClass.reset(new MatcherOpenCV(a, b));
    return loopClosingMatcherClass.get();
}
} // namespace putslam

// ---------------------------------------------------------------------------------------------
static int failures = 0;
#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            std::printf("CHECK failed line %d: %s\n", __LINE__, #c);  \
            ++failures;                                               \
        }                                                             \
    } while (0)

template <typename T> static bool rd(FILE *f, T *p, size_t n) { return std::fread(p, sizeof(T), n, f) == n; }

static cv::Mat descMat(const std::vector<unsigned char> &bytes, int rows)
{
    cv::Mat m(rows, 32, CV_8UC1);
    if (rows) std::memcpy(m.data, bytes.data(), (size_t)rows * 32);
    return m;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[8];
    if (!rd(f, hdr, 8)) return 2;
    const int n0 = hdr[0], n1 = hdr[1], nm = hdr[2], mode = hdr[3], nPairs = hdr[6];
    const uint64_t seedLC = (uint64_t)(uint32_t)hdr[4] | ((uint64_t)(uint32_t)hdr[5] << 32);
    std::vector<unsigned char> d0((size_t)n0 * 32), d1((size_t)n1 * 32);
    std::vector<Eigen::Vector3f> p0((size_t)n0), p1((size_t)n1);
    std::vector<cv::DMatch> ematches((size_t)nm);
    std::vector<int32_t> epairs((size_t)nPairs * 2);
    Eigen::Matrix4f epose;
    double eratio = 0;
    bool ok = rd(f, d0.data(), d0.size()) && rd(f, d1.data(), d1.size()) && rd(f, (float *)p0.data(), (size_t)n0 * 3) &&
              rd(f, (float *)p1.data(), (size_t)n1 * 3) && rd(f, (char *)ematches.data(), (size_t)nm * 16) &&
              rd(f, epairs.data(), epairs.size()) && rd(f, epose.data(), 16) && rd(f, &eratio, 1);
    int32_t vh[4];
    ok = ok && rd(f, vh, 4);
    const int F = vh[0], nV = vh[1];
    const uint64_t seedVO = (uint64_t)(uint32_t)vh[2] | ((uint64_t)(uint32_t)vh[3] << 32);
    std::vector<cv::Mat> fdesc;
    std::vector<std::vector<Eigen::Vector3f>> fpts;
    for (int k = 0; ok && k < F; ++k) {
        std::vector<unsigned char> d((size_t)nV * 32);
        std::vector<Eigen::Vector3f> p((size_t)nV);
        ok = rd(f, d.data(), d.size()) && rd(f, (float *)p.data(), (size_t)nV * 3);
        fdesc.push_back(descMat(d, nV));
        fpts.push_back(p);
    }
    std::vector<Eigen::Matrix4f> eposes((size_t)(F > 0 ? F - 1 : 0));
    for (auto &T : eposes) ok = ok && rd(f, T.data(), 16);
    std::fclose(f);
    if (!ok) {
        std::printf("short case file\n");
        return 2;
    }

    cv::Mat K(3, 3, CV_32FC1);
    const float Kv[9] = {517.3f, 0.0f, 318.6f, 0.0f, 516.5f, 255.3f, 0.0f, 0.0f, 1.0f};
    for (int i = 0; i < 9; ++i) K.at<float>(i / 3, i % 3) = Kv[i];
    RANSAC::parameters rp;
    rp.verbose = 0;
    rp.errorVersion = rp.errorVersionVO = 0;
    rp.errorVersionMap = mode;
    rp.inlierThresholdEuclidean = 0.04;
    rp.inlierThresholdReprojection = 2.0;
    rp.inlierThresholdMahalanobis = 0.0002;
    rp.minimalInlierRatioThreshold = 0.2;
    rp.minimalNumberOfMatches = 15;
    rp.usedPairs = 3;
    rp.iterationCount = 0;

    // (1) + (2a): reference-named factories and classes live next to the drop-in; performMatching == oracle
    putslam::Matcher *vo = putslam::createMatcherOpenCV();
    putslam::Matcher *lc = putslam::createloopClosingMatcherOpenCV("", "");
    CHECK(vo->getName() == "OpenCV Matcher" && lc->getName() == "OpenCVMatcher");
    cv::Mat A = descMat(d0, n0), B = descMat(d1, n1);
    std::vector<cv::DMatch> got = static_cast<MatcherOpenCV *>(vo)->publicPerformMatching(A, B);
    CHECK((int)got.size() == nm && std::memcmp(got.data(), ematches.data(), (size_t)nm * 16) == 0);

    // (2b) + (3): loop closure with the reference's argument types on a second thread while the VO matcher runs
    vo->matcherParameters.RANSACParams = rp;
    vo->matcherParameters.cameraMatrixMat = K;
    vo->frameDescriptors = &fdesc;
    vo->framePoints = &fpts;
    vo->syncParameters();
    vo->setSampleSeed(seedVO);
    RANSAC::parameters rlc = rp; // putslammatcherOpenCVParametersLC.xml:30
    rlc.minimalInlierRatioThreshold = 0.15;
    rlc.minimalNumberOfMatches = 10;
    lc->matcherParameters.RANSACParams = rlc;
    lc->matcherParameters.cameraMatrixMat = K;
    lc->syncParameters();
    lc->setSampleSeed(seedLC);

    std::vector<putslam::MapFeature> sets[2];
    int frames[2] = {7, 9};
    for (int s = 0; s < 2; ++s) {
        const int n = s ? n1 : n0;
        const std::vector<unsigned char> &d = s ? d1 : d0;
        const std::vector<Eigen::Vector3f> &p = s ? p1 : p0;
        for (int k = 0; k < n; ++k) {
            putslam::MapFeature mf;
            mf.id = (unsigned)(1000 * s + k);
            cv::Mat row(1, 32, CV_8UC1);
            std::memcpy(row.data, &d[(size_t)k * 32], 32);
            mf.descriptors[(unsigned)frames[s]] = putslam::ExtendedDescriptor(
                cv::Point2f(1.0f + k, 2.0f), cv::Point2f(10.0f + k, 20.0f + s),
                putslam::Vec3(p[(size_t)k][0], p[(size_t)k][1], p[(size_t)k][2]), row, 0, 1.0);
            // a view from another pose with different content: must not be the one that is read
            mf.descriptors[(unsigned)frames[s] + 100] = putslam::ExtendedDescriptor(
                cv::Point2f(), cv::Point2f(), putslam::Vec3(9, 9, 9), cv::Mat(1, 32, CV_8UC1), 3, 2.0);
            sets[s].push_back(mf);
        }
    }
    std::vector<std::pair<int, int>> pairs;
    Eigen::Matrix4f Tlc;
    double ratioLC = -5;
    std::thread th([&] { ratioLC = lc->matchFeatureLoopClosure(sets, frames, pairs, Tlc); });
    std::vector<Eigen::Matrix4f> poses;
    putslam::SensorFrame sf;
    sf.index = 0;
    vo->detectInitFeatures(sf);
    for (int k = 1; k < F; ++k) {
        sf.index = k;
        Eigen::Matrix4f T;
        std::vector<cv::DMatch> inl;
        double r = vo->runVO(sf, T, inl);
        CHECK(r > 0.2 && !inl.empty());
        poses.push_back(T);
    }
    th.join();
    CHECK(ratioLC == eratio);
    CHECK((int)pairs.size() == nPairs);
    bool samePairs = (int)pairs.size() == nPairs;
    for (int i = 0; samePairs && i < nPairs; ++i)
        samePairs = pairs[(size_t)i].first == epairs[(size_t)2 * i] && pairs[(size_t)i].second == epairs[(size_t)2 * i + 1];
    CHECK(samePairs);
    CHECK(std::memcmp(Tlc.data(), epose.data(), 64) == 0);
    CHECK(sets[0][3].u == 13.0 && sets[1][3].v == 21.0); // u, v taken from the analysed pose's view (:821-822)
    CHECK(poses.size() == eposes.size());
    for (size_t k = 0; k < poses.size() && k < eposes.size(); ++k) CHECK(std::memcmp(poses[k].data(), eposes[k].data(), 64) == 0);
    CHECK(vo->getNumberOfFeatures() == nV);

    // fewer than 10 features on either side -> 0, pairedFeatures untouched (matcher.cpp:830-834)
    std::vector<putslam::MapFeature> small[2] = {sets[0], std::vector<putslam::MapFeature>(sets[1].begin(), sets[1].begin() + 9)};
    std::vector<std::pair<int, int>> keep(3, std::make_pair(4, 2));
    CHECK(lc->matchFeatureLoopClosure(small, frames, keep, Tlc) == 0.0 && keep.size() == 3);
    // the matcher finds nothing (float descriptors are outside the path) -> -1.0 (matcher.cpp:838-839)
    {
        putslam_hip::FrameMatcherHIP hm;
        cv::Mat wide0(n0, 64, CV_8UC1), wide1(n1, 64, CV_8UC1);
        std::vector<cv::DMatch> inl;
        CHECK(hm.matchFeatureLoopClosure(wide0, p0, wide1, p1, Tlc, inl) == -1.0);
    }
    // matchXYZ with the reference's argument types == the FrameMatcher call it unpacks into (which the oracle checks in
    // tests/cpp/test_dropin.cpp): the previous set's features as the "map", the other set as the current pose
    {
        std::vector<putslam::MapFeature> mapF;
        std::vector<putslam_hip::FrameMatcher::MapFeatureXYZ> xyz;
        std::vector<int> viewOf;
        for (int k = 0; k < n0; ++k) {
            putslam::MapFeature mf = sets[0][(size_t)k];
            mf.position = putslam::Vec3(p0[(size_t)k][0], p0[(size_t)k][1], p0[(size_t)k][2]);
            auto &ext = mf.descriptors[(unsigned)frames[0]];
            ext.octave = k % 3;
            ext.detDist = std::sqrt(mf.position.x() * mf.position.x() + mf.position.y() * mf.position.y() +
                                    mf.position.z() * mf.position.z());
            mapF.push_back(mf);
            viewOf.push_back(frames[0]);
            putslam_hip::FrameMatcher::MapFeatureXYZ x;
            x.id = mf.id;
            x.position[0] = mf.position.x(); x.position[1] = mf.position.y(); x.position[2] = mf.position.z();
            x.descriptor = ext.descriptor;
            x.octave = ext.octave;
            x.detDist = ext.detDist;
            xyz.push_back(x);
        }
        std::vector<cv::KeyPoint> kps((size_t)n0);
        std::vector<double> det((size_t)n0);
        std::vector<int> oct((size_t)n0);
        std::vector<cv::Point2f> und((size_t)n0), dis((size_t)n0);
        std::vector<Eigen::Vector3f> curF = p0; // same positions, descriptors of the other frame where they correspond
        for (int k = 0; k < n0; ++k) {
            kps[(size_t)k].octave = oct[(size_t)k] = k % 3;
            det[(size_t)k] = std::sqrt((double)p0[(size_t)k][0] * p0[(size_t)k][0] + (double)p0[(size_t)k][1] * p0[(size_t)k][1] +
                                       (double)p0[(size_t)k][2] * p0[(size_t)k][2]);
            und[(size_t)k] = cv::Point2f(3.0f * k, 1.0f);
            dis[(size_t)k] = cv::Point2f(3.0f * k + 0.5f, 1.5f);
        }
        putslam_hip::FrameMatcherHIP h1, h2;
        for (putslam_hip::FrameMatcherHIP *h : {&h1, &h2}) {
            h->matcherParameters.RANSACParams = rlc;
            h->matcherParameters.cameraMatrixMat = K;
            h->setSampleSeed(77);
        }
        std::vector<putslam::MapFeature> found;
        Eigen::Matrix4f T1, T2;
        std::vector<cv::DMatch> inl2;
        const double r1 = putslam_hip::matchXYZ(h1, mapF, 42, found, T1, A, curF, kps, det, und, dis, viewOf, 1);
        const double r2 = h2.matchXYZ(xyz, A, curF, oct, det, T2, inl2, 1);
        CHECK(r1 == r2 && r1 > 0.5 && found.size() == inl2.size() && std::memcmp(T1.data(), T2.data(), 64) == 0);
        bool conv = found.size() == inl2.size();
        for (size_t i = 0; conv && i < found.size(); ++i) {
            const int mapId = inl2[i].queryIdx, cur = inl2[i].trainIdx;
            const putslam::MapFeature &mf = found[i];
            conv = mf.id == mapF[(size_t)mapId].id && mf.u == und[(size_t)cur].x && mf.posesIds.size() == 1 &&
                   mf.posesIds[0] == 42u && mf.descriptors.count(42u) == 1 &&
                   mf.descriptors.at(42u).octave == kps[(size_t)cur].octave &&
                   mf.position.z() == (double)curF[(size_t)cur][2] &&
                   std::memcmp(mf.descriptors.at(42u).descriptor.data, A.data + (size_t)cur * A.step, 32) == 0;
        }
        CHECK(conv);
    }
    if (failures == 0) std::printf("all checks passed\n");
    return failures ? 1 : 0;
}
