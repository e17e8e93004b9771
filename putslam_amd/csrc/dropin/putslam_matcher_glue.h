// putslam_matcher_glue.h -- the Matcher plugin's hot-path entry points with the REFERENCE'S OWN ARGUMENT TYPES.
//
// In a PUTSLAM build the reference keeps its class putslam::Matcher / ::MatcherOpenCV (detect / describe / track are
// image-domain OpenCV stages outside the path).  Its public methods
//     double match  (const SensorFrame&, Eigen::Matrix4f&, std::vector<cv::DMatch>&)            matcher.h:125-127
//     double runVO  (const SensorFrame&, Eigen::Matrix4f&, std::vector<cv::DMatch>&)            matcher.h:115-117
//     void   detectInitFeatures(const SensorFrame&)                                             matcher.h:97
//     double matchXYZ(std::vector<MapFeature>, int, std::vector<MapFeature>&, Eigen::Matrix4f&,
//                     bool, std::vector<int>, int)                                              matcher.h:131-136
//     double matchFeatureLoopClosure(std::vector<MapFeature>[2], int[2],
//                     std::vector<std::pair<int,int>>&, Eigen::Matrix4f&)                       matcher.h:139-140
// keep their signatures; their bodies call the templates below (INTEGRATION.md section 2 shows each body).  The
// templates are generic in the reference's data types (SensorFrame, MapFeature, ExtendedDescriptor, cv::KeyPoint):
// included from matcher.cpp they are instantiated with the real ones, tests/cpp/test_reference_shaped.cpp
// instantiates them with types of the same shape.  Header-only; links against libputslam_dropin.so.
#pragma once

#include <cstring>
#include <utility>
#include <vector>

#include "putslam_dropin.h"

namespace putslam_hip {

// Matcher::detectInitFeatures (matcher.cpp:17-64) / Matcher::match = runVO (matcher.cpp:452-516) from the point where
// the frame has been detected, described, undistorted and back-projected.  `frontEnd(sensorData, descriptors,
// features3D)` is the reference's own detect / describe / removeImageDistortion / keypoints2Dto3D sequence
// (matcher.cpp:457-467,474-480), injected by the caller; everything after it runs on the GPU.
template <class SensorFrameT, class FrontEnd>
void detectInitFeatures(FrameMatcher &hot, const SensorFrameT &sensorData, FrontEnd &&frontEnd)
{
    cv::Mat descriptors;
    std::vector<Eigen::Vector3f> features3D;
    frontEnd(sensorData, descriptors, features3D);
    hot.detectInitFeatures(descriptors, features3D);
}

template <class SensorFrameT, class FrontEnd>
double match(FrameMatcher &hot, const SensorFrameT &sensorData, FrontEnd &&frontEnd, Eigen::Matrix4f &estimatedTransformation,
             std::vector<cv::DMatch> &foundInlierMatches)
{
    cv::Mat descriptors;
    std::vector<Eigen::Vector3f> features3D;
    frontEnd(sensorData, descriptors, features3D);
    return hot.match(descriptors, features3D, estimatedTransformation, foundInlierMatches);
}

namespace detail {
inline cv::Mat stackRows32(size_t rows)
{
    return cv::Mat((int)rows, PUTSLAM_HIP_DESC_BYTES, CV_8UC1);
}
inline void putRow32(cv::Mat &dst, size_t row, const cv::Mat &src)
{
    if (!src.empty()) std::memcpy(dst.data + row * (size_t)dst.step, src.data, PUTSLAM_HIP_DESC_BYTES);
}
} // namespace detail

// Matcher::matchFeatureLoopClosure (matcher.cpp:802-861).  featureSets[i][k].descriptors[framesIds[i]] is the
// ExtendedDescriptor (point2DUndist, point3D, descriptor) of feature k seen from pose framesIds[i]; u, v of the
// features are updated like the reference does (:821-822).  Returns 0 when either set has fewer than 10 features
// (:830-834), -1.0 when the matcher finds nothing (:838-839), else RANSAC::pointInlierRatio(inliers, matches) with
// pairedFeatures = (queryIdx, trainIdx) of the RANSAC inliers under errorVersionMap (:843-858).
template <class MapFeatureT>
double matchFeatureLoopClosure(FrameMatcher &hot, std::vector<MapFeatureT> featureSets[2], int framesIds[2],
                               std::vector<std::pair<int, int>> &pairedFeatures, Eigen::Matrix4f &estimatedTransformation)
{
    // marshalling only: one 32-byte descriptor row and one float point per feature and side, in feature order
    cv::Mat descRows[2];
    std::vector<Eigen::Vector3f> xyz[2];
    for (int side = 0; side < 2; ++side) {
        std::vector<MapFeatureT> &feats = featureSets[side];
        const int view = framesIds[side];
        descRows[side] = detail::stackRows32(feats.size());
        xyz[side].resize(feats.size());
        for (size_t k = 0; k < feats.size(); ++k) {
            auto &seen = feats[k].descriptors[view];
            xyz[side][k] = Eigen::Vector3f((float)seen.point3D.x(), (float)seen.point3D.y(), (float)seen.point3D.z());
            feats[k].u = seen.point2DUndist.x; // (the reference refreshes u, v from the chosen view, :821-822)
            feats[k].v = seen.point2DUndist.y;
            detail::putRow32(descRows[side], k, seen.descriptor);
        }
    }
    if (xyz[0].size() < 10 || xyz[1].size() < 10) return 0;
    std::vector<cv::DMatch> inlierMatches;
    const double ratio = hot.matchFeatureLoopClosure(descRows[0], xyz[0], descRows[1], xyz[1], estimatedTransformation, inlierMatches);
    if (ratio == -1.0) return -1.0;
    pairedFeatures.clear();
    for (auto &m : inlierMatches) pairedFeatures.push_back(std::make_pair(m.queryIdx, m.trainIdx));
    return ratio;
}

// Matcher::matchXYZ, the private overload all public ones forward to (matcher.cpp:606-798).  mapFeatures[j] is matched
// through the ExtendedDescriptor of view frameIds[j] (or its first view, :672-676); the current pose is given by its
// descriptors, 3-D points, key points (octave) and detection distances; prevFeaturesUndistorted / Distorted are the
// reference's members of the same name, read when the inliers are converted back to MapFeatures (:770-792).
template <class MapFeatureT, class KeyPointT>
double matchXYZ(FrameMatcher &hot, std::vector<MapFeatureT> mapFeatures, int sensorPoseId,
                std::vector<MapFeatureT> &foundInlierMapFeatures, Eigen::Matrix4f &estimatedTransformation,
                cv::Mat currentPoseDescriptors, std::vector<Eigen::Vector3f> &currentPoseFeatures3D,
                std::vector<KeyPointT> &currentPoseKeyPoints, std::vector<double> &currentPoseDetDists,
                const std::vector<cv::Point2f> &prevFeaturesUndistorted, const std::vector<cv::Point2f> &prevFeaturesDistorted,
                std::vector<int> frameIds = std::vector<int>(), int computationNumber = 1)
{
    typedef typename std::remove_reference<decltype(mapFeatures[0].descriptors.begin()->second)>::type ExtendedDescriptorT;
    std::vector<FrameMatcher::MapFeatureXYZ> xyz(mapFeatures.size());
    for (size_t j = 0; j < mapFeatures.size(); ++j) {
        MapFeatureT &f = mapFeatures[j];
        ExtendedDescriptorT &ext = frameIds.size() > 0 ? f.descriptors[frameIds[j]] : f.descriptors.begin()->second;
        xyz[j].id = (unsigned int)f.id;
        xyz[j].position[0] = f.position.x();
        xyz[j].position[1] = f.position.y();
        xyz[j].position[2] = f.position.z();
        xyz[j].descriptor = ext.descriptor;
        xyz[j].octave = ext.octave;
        xyz[j].detDist = ext.detDist;
    }
    std::vector<int> octaves(currentPoseKeyPoints.size());
    for (size_t i = 0; i < octaves.size(); ++i) octaves[i] = currentPoseKeyPoints[i].octave;
    std::vector<cv::DMatch> inlierMatches;
    const double ratio = hot.matchXYZ(xyz, currentPoseDescriptors, currentPoseFeatures3D, octaves, currentPoseDetDists,
                                      estimatedTransformation, inlierMatches, computationNumber);
    if (ratio == -1.0) return -1.0;
    // inlier (map feature, current key point) pairs back to the caller's MapFeature type (:770-792): the map feature's id,
    // the key point's image position, 3-D point, descriptor row, octave and detection distance, seen from sensorPoseId
    typedef typename std::remove_reference<decltype(mapFeatures[0].position)>::type PositionT;
    foundInlierMapFeatures.clear();
    foundInlierMapFeatures.reserve(inlierMatches.size());
    for (const cv::DMatch &m : inlierMatches) {
        const size_t qi = (size_t)m.queryIdx, ti = (size_t)m.trainIdx;
        const Eigen::Vector3f &pt = currentPoseFeatures3D[ti];
        MapFeatureT out;
        out.id = mapFeatures[qi].id;
        out.u = prevFeaturesUndistorted[ti].x;
        out.v = prevFeaturesUndistorted[ti].y;
        out.position = PositionT((double)pt[0], (double)pt[1], (double)pt[2]);
        out.posesIds.push_back(sensorPoseId);
        cv::Mat row(1, PUTSLAM_HIP_DESC_BYTES, CV_8UC1, currentPoseDescriptors.data + ti * (size_t)currentPoseDescriptors.step);
        out.descriptors[sensorPoseId] = ExtendedDescriptorT(prevFeaturesUndistorted[ti], prevFeaturesDistorted[ti], out.position, row,
                                                            currentPoseKeyPoints[ti].octave, currentPoseDetDists[ti]);
        foundInlierMapFeatures.push_back(out);
    }
    return ratio;
}

} // namespace putslam_hip
